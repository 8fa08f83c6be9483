from .network_grid import NeRFNetwork, MLP  # noqa: F401
from .renderer import NeRFRenderer, get_default_torch_ngp_opt  # noqa: F401
