"""Per-scatter output record, same attributes as the reference's python/rayInterfaces.py:4-20.

Three extra attributes are filled by this implementation and absent from the reference: `orientation`
(+1 ray going into the surface, -1 out of it; the reference folds it into the sign of surface_index, which is lost
for surface 0), `surface_number` (0-based index of the surface hit) and `is_rayleigh`.
"""


class rayInterfaces(object):
    def __init__(self):
        self.incoming_ray = None
        self.reflected_ray = None
        self.refracted_ray = None
        self.intersection_point = None
        self.surface_normal = None
        self.ray_index = None
        self.surface_index = None
        self.distance_traveled = None
        self.n_incident = None
        self.n_transmitted = None
        self.bulkabs_incident = None
        self.bulkabs_transmitted = None
        self.rayleigh_incident = None
        self.rayleigh_transmitted = None
        self.orientation = None
        self.surface_number = None
        self.is_rayleigh = None
