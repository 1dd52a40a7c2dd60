#!/usr/bin/env python3
"""DI-VAE on PTB -- the reference's `ptb-utt.py` with the same flags (`--help` lists them; table and wiring in laed_amd/cli.py)."""
from laed_amd import cli

if __name__ == "__main__":
    cli.main("ptb-utt.py")
