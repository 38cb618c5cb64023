"""Drop-in for the reference's ``FrustumRegistration`` pybind module
(evaluation/frustum_reg/src/registration.cpp:190-213): ``solvePGivenK`` with the same keyword names,
plus the batched entry the HIP solver is built around."""
from .registration import __version__, solvePGivenK, solvePGivenK_batched  # noqa: F401
