#!/usr/bin/env python3
"""AE-ED: DI-VAE pretraining, then the encoder-decoder with frozen latent actions, on Stanford multi-domain -- the reference's `stanford-ae.py` with the same flags (`--help` lists them; table and wiring in laed_amd/cli.py)."""
from laed_amd import cli

if __name__ == "__main__":
    cli.main("stanford-ae.py")
