"""bench.py under a watchdog: dumps the Python stacks and exits when it has not finished in N seconds."""
import faulthandler
import runpy
import sys

faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
