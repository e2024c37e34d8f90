"""MI355X-native spherical sliced-Wasserstein registration loss (+ Chamfer baseline).

Importable as `importlib.import_module("sphere-homeomorphic-wasserstein-distance-for-point-cloud-registration_amd")`
or through the top-level alias module `shw_amd`."""
from . import _lib, dist
from .chamfer import chamfer_distance, chamfer_pair_losses
from .esw import esw_slice_sums, max_sliced_wasserstein_distance, rand_projections, sliced_wasserstein_distance
from .modules import (ChamferCriterion, GraphedAscent, GraphedStep, SlicedSphereW, SSWCriterion, max_cos_disimilarity_wassersten_distance,
                      max_spherical_wassersten_distance, max_spherical_wassersten_distance_fast)
from .sinkhorn import log_N_Sinkhorn_Distance_Loss, log_Sinkhorn_Distance_Loss, sinkhorn_pair_costs
from .ssw import (binary_search_circle, draw_directions, emd1D_circle, stiefel_frames, sliced_cost, sliced_wasserstein_sphere, sliced_wasserstein_sphere_fast,
                  ssw_pair_losses)

__all__ = ["_lib", "dist", "binary_search_circle", "emd1D_circle", "ChamferCriterion", "GraphedAscent", "GraphedStep", "SlicedSphereW", "SSWCriterion",
           "max_spherical_wassersten_distance", "max_spherical_wassersten_distance_fast",
           "max_cos_disimilarity_wassersten_distance", "chamfer_distance", "chamfer_pair_losses", "draw_directions", "log_N_Sinkhorn_Distance_Loss", "log_Sinkhorn_Distance_Loss", "sinkhorn_pair_costs", "stiefel_frames", "esw_slice_sums", "rand_projections", "sliced_wasserstein_distance", "sliced_cost", "sliced_wasserstein_sphere", "sliced_wasserstein_sphere_fast",
           "ssw_pair_losses"]
