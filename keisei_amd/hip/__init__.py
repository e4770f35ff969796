"""Python drivers of the HIP kernels (GPU-only code paths; nothing here runs on CPU tensors)."""
