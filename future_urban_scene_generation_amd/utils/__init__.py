"""Device-side replacements for hot helpers of the reference's ``utils`` package (only ``pnp_utils`` so far)."""
