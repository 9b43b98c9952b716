"""cProfile of the host side of the training step (where do the ~6 ms of Python per step go?)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline']
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(30)
