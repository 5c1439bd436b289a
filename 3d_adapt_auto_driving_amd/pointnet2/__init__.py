"""Host-side mirror of the reference's pointnet2_lib/pointnet2 Python package."""
