"""
torchdrivesim_amd -- MI355X (gfx950) implementation of the torchdrivesim hot path
(Simulator.step -> render_egocentric -> compute_collision / compute_offroad) behind the reference's plugin surface.
"""
__version__ = '0.1.0'
