#!/usr/bin/env python3
"""Runs __graft_entry__.smoke() (one small seg+flow call on cuda:0 checked against the oracle) and prints the outcome; a file instead of
`python -c` because tools/gpu_session.sh splits its step strings on blanks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.smoke()
print("smoke ok")
