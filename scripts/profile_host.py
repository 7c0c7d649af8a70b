"""cProfile of the main thread over N pipeline steps (prefetch mode, synthetic bench video): where the host
time of a step goes (Python bookkeeping vs waiting on the device)."""
import cProfile
import pstats
import sys
sys.path.insert(0, '.')
sys.argv = ['bench.py', '--steps', '300', '--warmup', '30', '--no-cpu-baseline']
import runpy
prof = cProfile.Profile()
prof.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
prof.disable()
st = pstats.Stats(prof)
st.sort_stats('tottime').print_stats(28)
