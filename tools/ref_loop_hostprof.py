"""cProfile of the reference's unchanged loop on the HIP model AFTER warm-up: where the host time of an eager iteration goes
(python tools/ref_loop_hostprof.py [steps] [rows])."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 45
sys.argv = [sys.argv[0], '5']
import runpy  # noqa: E402
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_loop_time.py'), run_name='__main__')
import torch  # noqa: E402
it, batches = ns['iteration'], ns['batches']
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    it(batches[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(rows)
