"""Stand-in package for torch_geometric (authoring container only; see ../README.md)."""
