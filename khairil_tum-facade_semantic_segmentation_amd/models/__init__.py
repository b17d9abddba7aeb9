"""Model surface mirrored from the reference's models/ directory (hot-path files only)."""
