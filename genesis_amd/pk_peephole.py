"""Assembly post-pass of the build (genesis_amd/build.py): no packed-fp32 instruction may take, in its LOW lane, its first
vector-register source from the low register of a pair and its second vector-register source from the HIGH register.

Measured on MI355X (tools/probe/pkfma_probe.hip, DESIGN.md finding 48): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with that
selection return a wrong low-lane result -- the HIGH register reads as 0 -- about 6e-5 of the time WHILE OTHER WORK SHARES THE GPU
(another process, or another stream of the same process), never when the kernel runs alone.  Forms that fail: op_sel = [0,1,..]
on two vector sources; op_sel = [1,0,1] / [0,1,1] of v_pk_fma_f32 when source 0 / 1 is a SCALAR pair (the vector sources are
then 1 / 0 and 2: low, HIGH).  Forms that are exact over > 1e12 evaluations each: every other low-lane selection ([0,0], [1,0],
[1,1], [0,0,1], [1,0,1] on vector sources), anything in the high lane, scalar sources with any selection next to ONE vector
source.  hipcc emits the failing forms wherever the SLP vectoriser folds a swapped pair into an operand selection (224 instructions
of this library).

The pass: (1) the three operations commute in sources 0 and 1 -- exchange them, and bits 0 / 1 of op_sel, op_sel_hi, neg_lo,
neg_hi with them: [0,1] becomes the exact form [1,0], not a bit of any result changes; (2) where that does not help (the third
source is involved), the instruction is split into its two scalar halves (v_fma_f32 / v_mul_f32 / v_add_f32 on the selected
registers, ordered so that the first does not overwrite a source of the second).  `count_bad()` is the verifier the build runs on
the final disassembly."""
import re

_PK = re.compile(r'^(\s*)(v_pk_(?:fma|mul|add)_f32)\s+([^/;]*?)\s*((?://|;).*)?$')
_MOD = re.compile(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]')
_PAIR = re.compile(r'^([vs])\[(\d+):(\d+)\]$')
_SCALAR_OP = {'v_pk_fma_f32': 'v_fma_f32', 'v_pk_mul_f32': 'v_mul_f32_e64', 'v_pk_add_f32': 'v_add_f32_e64'}


def _split_operands(s):
    """'v[0:1], v[2:3], v[4:5] op_sel:[0,1,0]' -> (['v[0:1]', 'v[2:3]', 'v[4:5]'], {'op_sel': [0,1,0]}, leftover text)"""
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


def _parse(line):
    m = _PK.match(line)
    if not m:
        return None
    ops, mods = _split_operands(m.group(3))
    nsrc = 3 if m.group(2) == 'v_pk_fma_f32' else 2
    if len(ops) != 1 + nsrc:
        return None
    full = {}
    for name, default in (('op_sel', 0), ('op_sel_hi', 1), ('neg_lo', 0), ('neg_hi', 0)):
        bits = list(mods.get(name, []))
        full[name] = bits + [default] * (nsrc - len(bits))
    return m.group(1), m.group(2), ops, full, mods, m.group(4) or ''


def _low_lane_bad(ops, op_sel):
    vec = [op_sel[i] for i in range(len(ops) - 1) if ops[1 + i].startswith('v')]
    return len(vec) >= 2 and vec[0] == 0 and vec[1] == 1


def is_bad(line):
    p = _parse(line)
    return bool(p) and _low_lane_bad(p[2], p[3]['op_sel'])


def _emit(indent, op, ops, full, given, comment):
    out = '%s%s %s' % (indent, op, ', '.join(ops))
    for name, default in (('op_sel', 0), ('op_sel_hi', 1), ('neg_lo', 0), ('neg_hi', 0)):
        if name in given or any(b != default for b in full[name]):
            out += ' %s:[%s]' % (name, ','.join(str(b) for b in full[name]))
    return out + ((' ' + comment) if comment else '')


def _half(operand, hi):
    m = _PAIR.match(operand)
    if not m or int(m.group(3)) != int(m.group(2)) + 1:
        raise ValueError('pk_peephole: cannot split operand %r' % operand)
    return '%s%d' % (m.group(1), int(m.group(2)) + (1 if hi else 0))


def fix_line(line):
    """The replacement for an affected instruction: a list of one (sources exchanged) or two (split) lines; None: not affected."""
    p = _parse(line)
    if not p or not _low_lane_bad(p[2], p[3]['op_sel']):
        return None
    indent, op, ops, full, given, comment = p
    if 'clamp' in line:
        raise ValueError('pk_peephole: clamp on an affected instruction: ' + line.strip())
    # (1) sources 0 and 1 exchanged
    sw_ops = [ops[0], ops[2], ops[1]] + ops[3:]
    sw = {k: [v[1], v[0]] + v[2:] for k, v in full.items()}
    if not _low_lane_bad(sw_ops, sw['op_sel']):
        return [_emit(indent, op, sw_ops, sw, given, comment)]
    # (2) the two lanes as scalar instructions
    nsrc = len(ops) - 1
    lanes = []
    for hi in (0, 1):
        sel = full['op_sel_hi'] if hi else full['op_sel']
        neg = full['neg_hi'] if hi else full['neg_lo']
        srcs = [('-' if neg[i] else '') + _half(ops[1 + i], sel[i]) for i in range(nsrc)]
        lanes.append((_half(ops[0], hi), srcs))
    (dlo, slo), (dhi, shi) = lanes
    reads = lambda srcs: set(x.lstrip('-') for x in srcs)     # noqa: E731
    if dlo not in reads(shi):
        order = [(dlo, slo), (dhi, shi)]
    elif dhi not in reads(slo):
        order = [(dhi, shi), (dlo, slo)]
    else:
        raise ValueError('pk_peephole: the halves of %r overwrite each other\'s sources' % line.strip())
    return ['%s%s %s, %s%s' % (indent, _SCALAR_OP[op], d, ', '.join(srcs), ('  ; pk_peephole: half of ' + line.strip()) if k == 0 else '')
            for k, (d, srcs) in enumerate(order)]


def rewrite(text):
    """(new assembly text, number of instructions rewritten)"""
    n = 0
    out = []
    for line in text.split('\n'):
        f = fix_line(line)
        if f is not None:
            n += 1
            out.extend(f)
        else:
            out.append(line)
    return '\n'.join(out), n


def count_bad(text):
    return sum(1 for line in text.split('\n') if is_bad(line))
