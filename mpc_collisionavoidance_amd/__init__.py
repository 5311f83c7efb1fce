"""MI355X-native batched SQP-RTI solver for the USV collision-avoidance OCPs of
ivanacollg/MPC_CollisionAvoidance, behind the acados_template calling convention."""
from .acados_template import AcadosModel, AcadosOcp, AcadosOcpSolver, BatchOcpSolver  # noqa: F401
from . import usv_models, scenario  # noqa: F401

__all__ = ["AcadosModel", "AcadosOcp", "AcadosOcpSolver", "BatchOcpSolver", "usv_models", "scenario"]
