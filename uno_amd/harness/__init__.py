"""Own counterparts of the reference's callers of the hot path:
the Darcy U-NO model, the relative-L2 loss, the complex-modulus Adam and the (data-parallel)
training step.  They exist so the hot path can be driven and measured end to end; they are not a
re-implementation of the reference's training scripts."""
from .models import UNO, UNO_9, Uno3D_T20  # noqa: F401
from .optim import ComplexAdam  # noqa: F401
from .losses import lp_loss_rel_sum  # noqa: F401
from .mixed import MixedDarcyTrainer  # noqa: F401
from .train import DarcyTrainer, GraphedStep, ns2d_rollout_loss, ns3d_loss, synthetic_darcy_batch  # noqa: F401
from . import workloads  # noqa: F401,E402
