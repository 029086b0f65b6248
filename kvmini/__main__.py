import sys

from kvmini.cli import main

if __name__ == "__main__":
    sys.exit(main())
