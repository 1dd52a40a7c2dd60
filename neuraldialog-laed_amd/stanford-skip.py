#!/usr/bin/env python3
"""ST-ED: DI-VST pretraining, then the encoder-decoder, on Stanford multi-domain -- the reference's `stanford-skip.py` with the same flags (`--help` lists them; table and wiring in laed_amd/cli.py)."""
from laed_amd import cli

if __name__ == "__main__":
    cli.main("stanford-skip.py")
