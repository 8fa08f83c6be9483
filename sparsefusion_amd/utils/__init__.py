"""Drop-ins for the reference's `utils/` modules that sit on the hot path's boundary:
the EFT feature renderer (ray samplers + CustomImplicitRenderer + LightFieldRaymarcher) and the loss / metric glue."""
