"""Drop-in mirror of the reference package `models` for the stage-1 hot path (HIP backend)."""
