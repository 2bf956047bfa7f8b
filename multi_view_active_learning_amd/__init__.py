"""multi_view_active_learning_amd -- MI355X (gfx950) implementation of the data-parallel
hot path of facebookresearch/multi_view_active_learning: heat-map CNN forward, keypoint
decode, RANSAC-DLT triangulation, uncertainty / core-set scoring.  See DESIGN.md."""
__version__ = "0.1.0"
