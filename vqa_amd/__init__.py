"""vqa_amd: MI355X-native backend for the hot path of Qulacs-Osaka/VQA --
batched circuit evaluation and parameter-shift gradients behind the QNN
target objects and the optimizers' gradient oracle."""
from .circuit import GateList, CZ, CNOT, H, X, Y, Z, RX, RY, RZ  # noqa: F401
from .observable import Observable, z_observable  # noqa: F401
from .learning_circuit import LearningCircuit, State  # noqa: F401
from .qml_utils import (  # noqa: F401
    create_ibm_HEA_ansatz,
    create_jerbi_ansatz,
    create_ry_rz_cz_ansatz,
    create_ry_cz_ansatz,
)
from .qnn_base import myQNN_base, myQNN_regression  # noqa: F401
from .qnn_classifier import myQNN_classifier  # noqa: F401
from .vqe_target import PauliSumVQE  # noqa: F401
from . import optimizers  # noqa: F401
from .sampling import sample_counts, sample_z_means  # noqa: F401

__version__ = "0.1.0"
