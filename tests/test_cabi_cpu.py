"""No-GPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/genesis_hip.h declares; argument validation fails loudly (no compute calls without a GPU);
the host-side model mirrors the reference's interface (state_dict layout, flags, no CPU fallback)."""
import ctypes
import os.path as osp
import re

import pytest
import torch

from genesis_amd import _lib
from oracle import v2_oracle as O

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(osp.join(REPO, 'include', 'genesis_hip.h')).read()
    declared = set(re.findall(r'\b(gx_\w+)\s*\(', header))
    decls = _lib.declarations()
    assert declared == set(decls), declared ^ set(decls)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gx_version() >= 1
    assert len(declared) >= 30


def test_argument_validation_needs_no_gpu():
    with pytest.raises(_lib.GenesisHipError, match='H in'):
        _lib.call('gx_conv3x3_fwd', None, None, None, 1, 3, 8, 0, 6, None, 0, None)
    with pytest.raises(_lib.GenesisHipError, match='powers of two'):
        _lib.call('gx_gn_relu_fwd', ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 8, 6, 6, 8,
                  ctypes.c_float(1e-5), ctypes.c_void_p(8), 8, 0, 0, None, 0, 0, 0, ctypes.c_void_p(8), ctypes.c_void_p(8), None)
    with pytest.raises(_lib.GenesisHipError, match='null pointer'):
        _lib.call('gx_mixture_fwd', None, None, 1, 8, 8, 3, ctypes.c_float(0.7), 1, None, None, None, None, None, 0, None)
    assert _lib.query('gx_conv3x3_ws_bytes', 32, 64, 64, 64, 64) >= 9 * 64 * 64 * 4
    assert _lib.query('gx_gn_relu_bwd_ws_bytes', 32, 64) == 32 * 64 * 3 * 4
    lib = _lib.load()
    assert lib.gx_profile_num_kernels() > 20 and lib.gx_profile_kernel_name(0) == b'tapconv_kernel<0>'


@pytest.mark.parametrize('over', [dict(K_steps=7, img_size=64, feat_dim=64), dict(K_steps=4, img_size=32, feat_dim=16),
                                  dict(K_steps=11, img_size=128, feat_dim=64),
                                  dict(K_steps=5, img_size=64, feat_dim=32, semiconv=False, autoreg_prior=False),
                                  dict(K_steps=5, img_size=64, feat_dim=64, kernel='epanechnikov')])
def test_state_dict_layout_matches_reference_contract(over):
    """Keys / shapes / dtypes of the product module == the reference's (oracle.param_shapes restates
    SURVEY.md Appendix B and is itself checked against the imported reference in test_oracle_vs_golden)."""
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    cfg = O.make_cfg(**over)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    sd = model.state_dict()
    want = O.param_shapes(cfg)
    assert list(sd.keys()) == list(want.keys())
    for k, (shape, _) in want.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sd['att_process.log_sigma'].dtype == O.default_log_sigma(cfg).dtype
    assert torch.equal(sd['att_process.log_sigma'], O.default_log_sigma(cfg))
    assert model.K_steps == cfg['K_steps']
    if over.get('feat_dim') == 64 and over['img_size'] == 64 and 'semiconv' not in over:
        assert len(sd) == 78 and sum(v.numel() for v in sd.values()) == 2729422


def test_flags_registered_like_the_reference():
    import genesis_amd.genesisv2_config  # noqa: F401
    from forge import flags
    for name, default in (('feat_dim', 64), ('kernel', 'gaussian'), ('semiconv', True), ('dynamic_K', False),
                          ('klm_loss', False), ('detach_mr_in_klm', True)):
        assert flags.FLAGS[name] == default


def test_forge_load_entry_point(tmp_path):
    """fet.load(model_config_path, cfg) -> load(cfg): the call train.py:152 makes."""
    from forge import experiment_tools as fet
    from genesis_amd.compat.attrdict import AttrDict
    path = osp.join(REPO, 'genesis_amd', 'genesisv2_config.py')
    cfg = AttrDict(O.make_cfg(K_steps=3, img_size=32, feat_dim=8), debug=False, multi_gpu=False)
    model = fet.load(path, cfg)
    assert type(model).__name__ == 'GenesisV2' and hasattr(model, 'sample') and hasattr(model, 'get_features')


def test_no_cpu_path():
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    model = G.load(AttrDict(dict(O.make_cfg(K_steps=3, img_size=32, feat_dim=8), debug=False, multi_gpu=False)))
    with pytest.raises(_lib.GenesisHipError, match='no CPU fallback'):
        model(torch.rand(1, 3, 32, 32))


def test_attrdict_semantics():
    from genesis_amd.compat.attrdict import AttrDict
    d = AttrDict(a=[1, 2], b={'c': 3})
    assert d.a == (1, 2) and d['a'] == [1, 2] and d.b.c == 3 and 'a' in d
    d['a'].append(3)
    assert d.a == (1, 2, 3)
    d.x = 5
    assert d['x'] == 5
    with pytest.raises(AttributeError):
        d.missing


def test_library_contexts():
    """gx_ctx_*: per-loop library state (no hidden state shared between independent users).  Contexts are
    created / switched / destroyed without a GPU; the deferred-reduction switch and the pending count are per context."""
    from genesis_amd import _lib
    lib = _lib.load()
    assert _lib.current_ctx() == 0
    a, b = int(lib.gx_ctx_create()), int(lib.gx_ctx_create())
    assert a > 0 and b > 0 and a != b
    try:
        _lib.make_current(a)
        assert _lib.current_ctx() == a
        _lib.call('gx_defer_enable', 1)
        _lib.make_current(b)
        assert _lib.query('gx_defer_pending') == 0
        # another thread starts in the default context, whatever this thread made current
        import threading
        seen = []
        t = threading.Thread(target=lambda: seen.append(_lib.current_ctx()))
        t.start(); t.join()
        assert seen == [0]
        with pytest.raises(_lib.GenesisHipError):
            _lib.make_current(31)                    # never created
    finally:
        _lib.make_current(0)
        _lib.call('gx_ctx_destroy', a)
        _lib.call('gx_ctx_destroy', b)
    with pytest.raises(_lib.GenesisHipError):
        _lib.make_current(a)                         # destroyed


def test_allreduce_entry_points_validate_before_touching_rccl():
    """gx_allreduce_* (the step's collective behind the C ABI, include/genesis_hip.h): the id size and the argument checks run
    before RCCL is resolved, so they can be exercised without a GPU."""
    import ctypes
    from genesis_amd import _lib
    lib = _lib.load()
    assert int(lib.gx_allreduce_unique_id_bytes()) == 128
    h = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(128)
    for rank, world in ((2, 2), (-1, 1), (0, 0)):
        assert lib.gx_allreduce_init(ident, 128, rank, world, ctypes.byref(h)) != 0
        assert 'rank' in _lib.last_error()
    assert lib.gx_allreduce_init(ident, 64, 0, 1, ctypes.byref(h)) != 0            # id buffer too small
    assert lib.gx_allreduce_unique_id(ident, 16) != 0
    assert lib.gx_allreduce_run(None, None, 0, None) != 0                            # null communicator
    assert lib.gx_allreduce_destroy(None) == 0
