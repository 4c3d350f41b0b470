from .builder import APPLICATION, NETWORK, REGRESSION, build_actor_critic, build_all, build_target_network  # noqa: F401
from .mlp import LinearMLP  # noqa: F401
from .pointnet import AugmentedObs, PointNet  # noqa: F401
from .visuomotor import Visuomotor  # noqa: F401
from .heads import TanhGaussianHead  # noqa: F401
from .actor_critic import ContinuousActor, ContinuousCritic  # noqa: F401
