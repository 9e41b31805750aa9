"""Import-only stub used when importing the reference in the build container
(utils/misc.py:16 does `import simplejson as json`). No arithmetic."""
from json import *  # noqa: F401,F403
