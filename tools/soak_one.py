import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_fuzz as t
seed, order = int(sys.argv[1]), sys.argv[2]
try:
    t._world(seed, order)
    print("OK", seed, order)
except AssertionError as e:
    print("MISMATCH", str(e)[:1500])
