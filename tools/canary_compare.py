#!/usr/bin/env python3
"""Entry-point shim kept at the path users of the flat-script layout expect:
`python tools/canary_compare.py ...` == `kvmini canary ...`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvmini.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(["canary"] + sys.argv[1:]))
