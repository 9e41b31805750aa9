"""sample() of GENESIS (models/genesis_config.py:345-425), MONet (models/monet_config.py:172-198) and BaselineVAE
(models/vae_config.py:89-96) against the reference's own sample() outputs on its recorded standard-normal draws
(tests/golden/make_golden_sample_models.py): the oracle restatements on the CPU, the HIP path on the GPU."""
import json
import os.path as osp

import numpy as np
import pytest
import torch

from genesis_amd import testing as T

GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), 'golden')
GEN_CASES = ['tiny', 'tiny_in', 'tiny_noprior', 'tiny_onestage', 'tiny_sym', 'tiny_train', 'cfg3']
MONET_CASES = ['tiny', 'tiny_k5', 'tiny_scope', 'cfg4']
VAE_CASES = ['tiny', 'tiny_bcast', 'cfg1']
DEV = 'cuda'
st = lambda l: torch.stack(list(l))  # noqa: E731


class SGold(object):
    def __init__(self, family, name):
        self.name = '%s_sample_%s' % (family, name)
        self.g = np.load(osp.join(GOLDEN, self.name + '.npz'), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.B = int(self.g['B'])
        self.K_arg = None if int(self.g['K_arg']) < 0 else int(self.g['K_arg'])
        self.train = bool(int(self.g['train_mode']))
        self.S = self.cfg['img_size']

    def t(self, key):
        return torch.from_numpy(self.g[key])

    def weights(self, template):
        assert list(template.keys()) == [str(k) for k in self.g['sd_keys']]
        assert [int(v.numel()) for v in template.values()] == [int(n) for n in self.g['sd_numel']]
        return T.formula_state_dict(template)

    def check(self, key, tensor, rtol, atol):
        full = 'out/' + key
        if full in self.g.files:
            np.testing.assert_allclose(tensor.detach().cpu().float().numpy(), self.g[full], rtol=rtol, atol=atol,
                                       err_msg='%s %s' % (self.name, key))
        else:
            T.check_summary(full, tensor, self.g, rtol, atol, self.name)


# ---------------------------------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize('case', GEN_CASES)
def test_oracle_genesis_sample(case):
    from oracle import genesis_oracle as GO
    gold = SGold('genesis', case)
    p = gold.weights(GO.template_state_dict(gold.cfg))
    eps_c = list(gold.t('eps_c').unbind(0)) if 'eps_c' in gold.g.files else None
    with torch.no_grad():
        img, x_k, log_m_k, log_s_k, zm_k, zc_k = GO.genesis_sample(p, gold.cfg, list(gold.t('eps_m').unbind(0)), eps_c,
                                                                   training=gold.train)
    gold.check('zm_k', st(zm_k), 2e-5, 2e-5)
    if zc_k is not None:
        gold.check('zc_k', st(zc_k), 2e-5, 2e-5)
    gold.check('img', img, 2e-5, 2e-5)
    gold.check('x_k', st(x_k), 2e-5, 2e-5)
    gold.check('log_m_k', st(log_m_k), 5e-5, 5e-5)
    gold.check('log_s_k', st(log_s_k), 5e-5, 5e-5)


@pytest.mark.parametrize('case', MONET_CASES)
def test_oracle_monet_sample(case):
    from oracle import monet_oracle as MO
    gold = SGold('monet', case)
    p = gold.weights(MO.template_state_dict(gold.cfg))
    with torch.no_grad():
        img, x_k, log_m_k = MO.monet_sample(p, gold.cfg, gold.t('eps'), gold.K_arg)
    gold.check('img', img, 2e-5, 2e-5)
    gold.check('x_k', st(x_k), 2e-5, 2e-5)
    gold.check('log_m_k', st(log_m_k), 5e-5, 5e-5)


@pytest.mark.parametrize('case', VAE_CASES)
def test_oracle_vae_sample(case):
    from oracle import vae_oracle as VO
    gold = SGold('vae', case)
    p = gold.weights(VO.template_state_dict(gold.cfg))
    with torch.no_grad():
        img = VO.vae_sample(p, gold.cfg, gold.t('eps'))
    gold.check('img', img, 2e-5, 2e-5)


# ---------------------------------------------------------------------------------------- HIP path (GPU)
def build(gold, mod):
    from genesis_amd.compat.attrdict import AttrDict
    cfg = AttrDict(dict(gold.cfg, debug=False, multi_gpu=False))
    torch.manual_seed(0)
    model = mod.load(cfg)
    model.load_state_dict(gold.weights(model.state_dict()))
    model = model.to(DEV)
    return model.train() if gold.train else model.eval()


@pytest.mark.gpu
@pytest.mark.parametrize('case', GEN_CASES)
def test_genesis_sample_vs_golden(case):
    import genesis_amd.genesis_config as G
    gold = SGold('genesis', case)
    model = build(gold, G)
    eps_c = gold.t('eps_c').to(DEV) if 'eps_c' in gold.g.files else None
    args = (gold.B,) if gold.K_arg is None else (gold.B, gold.K_arg)
    img, stats = model.sample(*args, eps_m=gold.t('eps_m').to(DEV), eps_c=eps_c)
    K = gold.cfg['K_steps']
    assert len(stats.x_k) == K and len(stats.log_m_k) == K and len(stats.log_s_k) == K + 1 and len(stats.mx_k) == K
    gold.check('zm_k', st(stats.zm_k), 1e-4, 2e-5)
    if eps_c is not None:
        gold.check('zc_k', st(stats.zc_k), 1e-4, 2e-5)
    gold.check('img', img, 1e-4, 2e-5)
    gold.check('x_k', st(stats.x_k), 1e-4, 2e-5)
    gold.check('log_m_k', st(stats.log_m_k), 1e-4, 2e-4)
    gold.check('log_s_k', st(stats.log_s_k), 1e-4, 2e-4)
    gold.check('mx_k', st(stats.mx_k), 1e-4, 2e-5)
    # the call train.py:463 makes (own draws): shapes, masks sum to one
    img2, st2 = model.sample(batch_size=8, K_steps=model.K_steps)
    assert img2.shape == (8, 3, gold.S, gold.S) and not img2.requires_grad
    assert float((torch.stack(st2.log_m_k, 4).exp().sum(4) - 1).abs().max()) < 1e-3


@pytest.mark.gpu
def test_genesis_sample_rejects_another_slot_count():
    import genesis_amd.genesis_config as G
    gold = SGold('genesis', 'tiny')
    model = build(gold, G)
    with pytest.raises(AssertionError):            # the reference asserts len(zm_k) == self.K_steps (:379)
        model.sample(2, gold.cfg['K_steps'] + 1)


@pytest.mark.gpu
@pytest.mark.parametrize('case', MONET_CASES)
def test_monet_sample_vs_golden(case):
    import genesis_amd.monet_config as G
    gold = SGold('monet', case)
    model = build(gold, G)
    img, stats = model.sample(gold.B, gold.K_arg, eps=gold.t('eps').to(DEV))
    K = gold.cfg['K_steps'] if gold.K_arg is None else gold.K_arg
    assert len(stats.x_k) == K and torch.equal(stats.gen_image, img)
    gold.check('img', img, 1e-4, 2e-5)
    gold.check('x_k', st(stats.x_k), 1e-4, 2e-5)
    gold.check('log_m_k', st(stats.log_m_k), 1e-4, 2e-4)
    gold.check('mx_k', st(stats.mx_k), 1e-4, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('case', VAE_CASES)
def test_vae_sample_vs_golden(case):
    import genesis_amd.vae_config as G
    gold = SGold('vae', case)
    model = build(gold, G)
    eps = gold.t('eps').to(DEV)
    img, stats = model.sample(gold.B, eps=eps)
    assert torch.equal(stats.z, eps)
    gold.check('img', img, 1e-4, 2e-5)
    img2, _ = model.sample(5, 3)                  # train.py:463 passes K_steps positionally or by keyword
    assert img2.shape == (5, 3, gold.S, gold.S)
