"""Import-only stub (utils/misc.py:22 imports tensorflow; nothing on the hot path uses it)."""
