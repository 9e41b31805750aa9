"""Assembly post-pass of the build (genesis_amd/build.py): no packed-fp32 instruction may take, in its LOW lane, source 0 from the
low register of its pair and source 1 from the HIGH one.

Measured on MI355X (tools/probe/pkfma_probe.hip, DESIGN.md finding 48): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with
op_sel = [0,1,..] return a wrong low-lane result -- source 1 reads as 0 -- about 6e-5 of the time WHILE OTHER WORK SHARES THE GPU
(another process, or another stream of the same process), never when the kernel runs alone; every other selection (op_sel [0,0],
[1,0], [1,1]; anything in the high lane) is exact over > 1e12 evaluations.  hipcc emits the form wherever the SLP vectoriser folds
a swapped pair into an operand selection (222 instructions of this library).  All three operations commute in sources 0 and 1,
so the pass swaps them -- and bits 0 / 1 of op_sel, op_sel_hi, neg_lo, neg_hi with them --, which turns [0,1] into the exact
form [1,0] without changing a bit of any result.  `count_bad()` is the verifier the build runs on the final disassembly."""
import re

_PK = re.compile(r'^(\s*)(v_pk_(?:fma|mul|add)_f32)\s+([^/;]*?)\s*((?://|;).*)?$')
_MOD = re.compile(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]')


def _split_operands(s):
    """'v[0:1], v[2:3], v[4:5] op_sel:[0,1,0]' -> (['v[0:1]', 'v[2:3]', 'v[4:5]'], {'op_sel': [0,1,0]})"""
    mods = {m.group(1): [int(b) for b in m.group(2).split(',')] for m in _MOD.finditer(s)}
    body = _MOD.sub('', s).strip()
    ops, depth, cur = [], 0, ''
    for ch in body:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            ops.append(cur.strip())
            cur = ''
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops, mods


def is_bad(line):
    m = _PK.match(line)
    if not m:
        return False
    _, mods = _split_operands(m.group(3))
    sel = mods.get('op_sel')
    return bool(sel) and len(sel) >= 2 and sel[0] == 0 and sel[1] == 1


def fix_line(line):
    """The instruction with sources 0 and 1 exchanged (None: not an affected instruction)."""
    m = _PK.match(line)
    if not m or not is_bad(line):
        return None
    indent, op, rest, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ''
    ops, mods = _split_operands(rest)
    nsrc = 3 if op == 'v_pk_fma_f32' else 2
    assert len(ops) == 1 + nsrc, line
    ops[1], ops[2] = ops[2], ops[1]
    # op_sel_hi defaults to all ones, the others to all zeros: only materialise what is not the default
    for name in ('op_sel', 'op_sel_hi', 'neg_lo', 'neg_hi'):
        if name in mods:
            bits = mods[name]
            bits[0], bits[1] = bits[1], bits[0]
    out = '%s%s %s' % (indent, op, ', '.join(ops))
    for name in ('op_sel', 'op_sel_hi', 'neg_lo', 'neg_hi'):
        if name in mods:
            out += ' %s:[%s]' % (name, ','.join(str(b) for b in mods[name]))
    return out + ((' ' + comment) if comment else '')


def rewrite(text):
    """(new assembly text, number of instructions rewritten)"""
    n = 0
    out = []
    for line in text.split('\n'):
        f = fix_line(line)
        if f is not None:
            n += 1
            out.append(f)
        else:
            out.append(line)
    return '\n'.join(out), n


def count_bad(text):
    return sum(1 for line in text.split('\n') if is_bad(line))
