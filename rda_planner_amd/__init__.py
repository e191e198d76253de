"""rda_planner_amd - MI355X-native RDA ADMM inner solver behind the reference's MPC / RDA_solver API."""
from .scenarios import car  # noqa: F401

__all__ = ["MPC", "RDA_solver", "car"]


def __getattr__(name):
    if name == "MPC":
        from .mpc import MPC
        return MPC
    if name == "RDA_solver":
        from .rda_solver import RDA_solver
        return RDA_solver
    raise AttributeError(name)
