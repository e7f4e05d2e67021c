#!/usr/bin/env python3
"""`python train.py <reference flags>` -- thin launcher for otgan_amd.train (the MI355X-native
replacement of the reference's train.py; same command line)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from otgan_amd.train import main  # noqa: E402

if __name__ == '__main__':
    main(self_launch=True)
