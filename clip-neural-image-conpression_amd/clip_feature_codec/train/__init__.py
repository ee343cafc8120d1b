"""Training step of the diffusion decoder on the MI355X kernels (the loop body of the reference's train/diffusion_train.py)."""
