"""Mirror of the reference subpackage of the same name (hot-path modules only)."""
