"""CLI entry point with the reference's location and usage:

    python cchess_alphazero/run.py self [--type mini|normal|distribute] [--gpu 0,1,...]
    python cchess_alphazero/run.py eval [--type ...]

The work is done by the MI355X engine (see manager.py / worker/self_play.py).
"""
import multiprocessing
import pathlib
import sys


def main():
    package_parent = str(pathlib.Path(__file__).resolve().parent.parent)
    if package_parent not in sys.path:
        sys.path.insert(0, package_parent)
    multiprocessing.set_start_method("spawn", force=True)      # one process per GPU is spawned, never forked
    from cchess_alphazero import manager
    return manager.start()


if __name__ == "__main__":
    main()
