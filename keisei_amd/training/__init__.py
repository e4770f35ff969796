"""Host-side mirror of the reference's ``keisei.training`` hot-path API (same names, arguments,
error behaviour), backed by hand-written HIP kernels for GPU tensors."""
