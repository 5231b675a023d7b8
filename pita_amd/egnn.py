"""Time-only EGNN backbone (mirror of pita/src/models/components/egnn.py:7-117: ``forward(t, xs)``,
``in_node_nf=1``); shares the HIP kernel with the temperature-conditioned variant."""
import torch

from .egnn_temp_conditioned import EGNN, EGNN_dynamics as _TempEGNN  # noqa: F401


class EGNN_dynamics(_TempEGNN):
    def __init__(self, n_particles, n_dimension, hidden_nf=64, act_fn=torch.nn.SiLU(), n_layers=4, recurrent=True,
                 attention=False, condition_time=True, tanh=False, agg="sum", energy_function=None, energy=False,
                 add_virtual=False, precision="bf16x3"):
        super().__init__(n_particles, n_dimension, hidden_nf=hidden_nf, act_fn=act_fn, n_layers=n_layers,
                         recurrent=recurrent, attention=attention, condition_time=condition_time, tanh=tanh, agg=agg,
                         energy=energy, add_virtual=add_virtual, condition_temperature=False, precision=precision)

    def forward(self, t, xs, beta=None):  # beta accepted and ignored so ScoreNet can wrap either variant
        return super().forward(t, xs, None)
