"""MI355X-native differentiable Gaussian-splat rasterizer (IGS hot path)."""
