"""Command line entry points (mirror of stardist/scripts, setup.py:155-160)."""
