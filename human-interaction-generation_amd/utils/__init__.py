"""Mirror of the part of codes/utils the sampling path needs after the denoiser (joint recovery)."""
