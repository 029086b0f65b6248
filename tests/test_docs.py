"""Documentation that is generated from the code stays in sync with it."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_reference_matches_the_parsers():
    """docs/CLI.md is the `--help` of every subcommand (scripts/gen_cli_docs.py): a flag added to a
    parser without regenerating the reference fails here."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_cli_docs.py"), "--check"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]


def test_every_subcommand_and_make_target_named_in_the_docs_exists():
    import re

    from kvmini.cli import COMMANDS

    targets = set(re.findall(r"^([a-zA-Z][\w-]*):", open(os.path.join(ROOT, "Makefile")).read(), re.M))
    for name in ("README.md", "docs/DEVELOPER_SETUP.md", "docs/LINTING.md", "docs/TROUBLESHOOTING.md", "docs/API.md"):
        text = open(os.path.join(ROOT, name)).read()
        for cmd in re.findall(r"`kvmini ([a-z][a-z-]*)", text):
            assert cmd in COMMANDS, (name, cmd)
        for tgt in re.findall(r"`make ([a-z][\w-]*)", text):
            assert tgt in targets, (name, tgt)
