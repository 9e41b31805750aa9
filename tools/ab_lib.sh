# A/B of two builds of the library on the canvas conv and the transposed-conv launches:  tools/ab_lib.sh <lib A> <lib B>
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L"
  GENESIS_HIP_LIB=$R/$L python $R/tools/c3p_check.py /tmp/$(basename $L).pt 2>&1 | grep "(224, 32, 72)"
  GENESIS_HIP_LIB=$R/$L python $R/tools/kq_time.py fwd:224:32 dgrad:224:32 fwd:224:16 dgrad:224:16 2>&1 | grep -v amdgpu
done
python - "$@" <<'PY'
import sys, torch, os
a, b = [torch.load('/tmp/%s.pt' % os.path.basename(p)) for p in sys.argv[1:3]]
print('canvas conv outputs bit-identical between the two builds:', all(torch.equal(a[k], b[k]) for k in a))
PY
