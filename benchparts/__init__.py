"""Pieces of bench.py (repo root): common helpers, the CPU baseline, the fan-out parent, the untimed extras and PGD blocks."""
