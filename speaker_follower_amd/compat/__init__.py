"""Directory of bare-name modules (`model`, `follower`, `speaker`, `env`) that re-export this package's
mirrors of the reference's tasks/R2R modules; see README.md.  `path()` is what goes on sys.path."""
import os


def path():
    return os.path.dirname(os.path.abspath(__file__))
