"""Debug helper: bench.main() with a timed all-thread traceback dump (hang diagnosis)."""
import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "75")), exit=True)
import bench
bench.main()
