"""Namespace of the import shims (see ../README.md): no D-Wave software is involved."""
