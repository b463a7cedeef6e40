"""``python cchess_alphazero/run.py self [--type mini|normal|distribute] [--gpu 0,1,...]`` -- same entry
point as the reference's cchess_alphazero/run.py."""
import multiprocessing as mp
import os
import sys

_PATH_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PATH_ not in sys.path:
    sys.path.insert(0, _PATH_)

if __name__ == "__main__":
    mp.set_start_method('spawn')
    sys.setrecursionlimit(10000)
    from cchess_alphazero import manager
    manager.start()
