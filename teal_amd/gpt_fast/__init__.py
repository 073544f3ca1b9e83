"""Decode harness around the sparse-GEMV hot path (gpt-fast-shaped Llama + generate.py flags)."""
