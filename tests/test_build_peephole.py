"""The build's assembly post-pass (genesis_amd/pk_peephole.py, DESIGN.md finding 48): packed-fp32 instructions whose LOW lane takes
its first vector-register source from the low and its second from the HIGH register of their pairs misread that register while
other work shares an MI355X; the pass exchanges the two commuting sources, or splits the instruction into its scalar halves.  CPU tests: the rewriting itself, and that no such instruction is left in the
device code of the built library."""
import os.path as osp

import pytest

from genesis_amd import build as B
from genesis_amd import pk_peephole as P


@pytest.mark.parametrize('line,fixed', [
    ('\tv_pk_fma_f32 v[60:61], v[84:85], v[82:83], v[60:61] op_sel:[0,1,0] op_sel_hi:[0,0,1]',
     ['\tv_pk_fma_f32 v[60:61], v[82:83], v[84:85], v[60:61] op_sel:[1,0,0] op_sel_hi:[0,0,1]']),
    ('\tv_pk_add_f32 v[4:5], v[4:5], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]// 0000002F5EC8: D3B25004 08020904',
     ['\tv_pk_add_f32 v[4:5], v[4:5], v[4:5] op_sel:[1,0] op_sel_hi:[0,1] // 0000002F5EC8: D3B25004 08020904']),
    ('\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]', ['\tv_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,0]']),
    ('\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[0,1,0]',
     ['\tv_pk_fma_f32 v[0:1], v[4:5], v[2:3], v[0:1] op_sel:[1,0,0] neg_lo:[0,1,0] neg_hi:[1,0,0]']),
    # a scalar pair among the sources: the vector sources are 1 and 2 (low, HIGH) -- exchanging 0 and 1 does not help: split
    ('\tv_pk_fma_f32 v[24:25], s[20:21], v[24:25], v[68:69] op_sel:[1,0,1]',
     ['\tv_fma_f32 v24, s21, v24, v69  ; pk_peephole: half of v_pk_fma_f32 v[24:25], s[20:21], v[24:25], v[68:69] op_sel:[1,0,1]',
      '\tv_fma_f32 v25, s21, v25, v69']),
    ('\tv_pk_fma_f32 v[24:25], v[24:25], s[20:21], v[68:69] op_sel:[0,1,1] neg_lo:[1,0,0]',
     ['\tv_fma_f32 v24, -v24, s21, v69  ; pk_peephole: half of v_pk_fma_f32 v[24:25], v[24:25], s[20:21], v[68:69] op_sel:[0,1,1] neg_lo:[1,0,0]',
      '\tv_fma_f32 v25, v25, s21, v69']),
    # the high half first where the low half would overwrite one of its sources
    ('\tv_pk_fma_f32 v[0:1], s[2:3], v[0:1], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,0,1]',
     ['\tv_fma_f32 v1, s3, v0, v7  ; pk_peephole: half of v_pk_fma_f32 v[0:1], s[2:3], v[0:1], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,0,1]',
      '\tv_fma_f32 v0, s2, v0, v7']),
    # not affected: left alone
    ('\tv_pk_mul_f32 v[4:5], v[4:5], v[26:27] op_sel:[1,0] op_sel_hi:[0,1]', None),
    ('\tv_pk_fma_f32 v[58:59], v[86:87], v[76:77], v[58:59] op_sel_hi:[0,1,1]', None),
    ('\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,1,0] op_sel_hi:[0,0,1]', None),
    ('\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]', None),
    ('\tv_pk_mul_f32 v[2:3], v[4:5], s[6:7] op_sel:[0,1]', None),
    ('\tv_pk_fma_f16 v0, v1, v2, v0 op_sel:[0,1,0]', None),
    ('\tv_fma_f32 v0, v1, v2, v0', None),
])
def test_rewrites_exactly_the_affected_operand_selection(line, fixed):
    assert P.fix_line(line) == fixed
    assert P.is_bad(line) == (fixed is not None)
    if fixed is not None:
        assert not any(P.is_bad(f) for f in fixed)
        text, n = P.rewrite('s_nop 0\n' + line + '\ns_endpgm')
        assert n == 1 and P.count_bad(text) == 0 and text.split('\n')[1:-1] == fixed


def test_refuses_what_it_cannot_split():
    with pytest.raises(ValueError):
        P.fix_line('\tv_pk_fma_f32 v[0:1], s[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]')


def test_no_affected_instruction_in_the_built_library():
    """Disassembles the device code of every object of the in-tree build (the objects libgenesis_hip.so is linked from)."""
    objs = [osp.join(B.CSRC, osp.splitext(s)[0] + '.o') for s in B.SOURCES]
    objs = [o for o in objs if osp.exists(o)]
    if not objs or not osp.exists(osp.join(B.LLVM_BIN, 'llvm-objdump')):
        pytest.skip('no in-tree objects / no llvm-objdump here')
    assert B.verify_objects(objs) == {}
