#!/usr/bin/env python3
"""DI-VST (skip-thought) on DailyDialog -- the reference's `dailydialog-utt-skip.py` with the same flags (`--help` lists them; table and wiring in laed_amd/cli.py)."""
from laed_amd import cli

if __name__ == "__main__":
    cli.main("dailydialog-utt-skip.py")
