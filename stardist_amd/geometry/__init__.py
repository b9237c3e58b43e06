"""Host-side mirror of the reference's stardist/geometry package (hot-path functions only)."""
from .geom2d import star_dist, dist_to_coord, polygons_to_label, polygons_to_label_coord, ray_angles
from .geom3d import star_dist3D, polyhedron_to_label, dist_to_coord3D
