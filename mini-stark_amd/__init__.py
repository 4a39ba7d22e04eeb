# see ../mini_stark_amd.py (import shim) — this directory is the package body.
