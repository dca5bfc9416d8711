"""Test stub: `audioread` is only used when AudioURL is set (scripts/Encoder.py:331); the test configs set none."""
