"""The reference's second peptide EGNN, ``EGNN_dynamics`` of pita/src/models/components/egnn_aldp.py:8-197 ("EGNN from
Equivariant FM": hidden 64 x 4 layers, no attention gate and no tanh bound by default, one-hot atom-type node features
with t and -- optionally -- beta appended), on the HIP kernels of the wide backbone.

Its layer is the E_GCL of egnn.py line for line (egnn_aldp.py:200-436 against egnn.py:108-346) and its forward differs
from ``EGNN_dynamics_AD2_cat`` only in the static feature table (its own methyl grouping for 22 atoms, :51-56) and in the
constructor's names and defaults, so it is that module with this class's signature: same parameter names and creation
order (``egnn.embedding``, ``egnn.embedding_out``, ``egnn.gcl_<l>...``: the reference's ``state_dict`` loads unchanged),
``forward(t, xs, beta) -> vel``, and -- through the base class -- ``edm``, ``sampler_run`` and ``jvp``.
"""
import numpy as np
import torch

from .egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat


class EGNN_dynamics(EGNN_dynamics_AD2_cat):
    def __init__(self, n_particles, n_dimension, hidden_nf=64, act_fn=torch.nn.SiLU(), n_layers=4, recurrent=True,
                 attention=False, condition_time=True, tanh=False, agg="sum", condition_temperature=False,
                 h_initial=None):
        if not condition_time:
            # (:134-137 appends t only when asked; every reference use conditions on time, and the kernels' node-feature
            # embedding always carries the t column)
            raise NotImplementedError("egnn_aldp.EGNN_dynamics(condition_time=False) is not built")
        if h_initial is None:
            h_initial = self._aldp_h_initial(n_particles)
        super().__init__(n_particles, n_dimension, hidden_nf=hidden_nf, act_fn=act_fn, n_layers=n_layers,
                         recurrent=recurrent, attention=attention, tanh=tanh, agg=agg,
                         condition_beta=condition_temperature, h_initial=h_initial)
        self._n_dimension = n_dimension  # the reference's attribute name (:25)
        self.condition_time, self.condition_temperature = condition_time, condition_temperature
        self.in_node_nf = int(self.h_initial.shape[-1]) + 1 + (1 if condition_temperature else 0)

    @staticmethod
    def _aldp_h_initial(n):
        """get_h_initial of egnn_aldp.py:51-78 for the particle counts that need no topology."""
        groups = {22: [([1, 2, 3], 2), ([19, 20, 21], 20), ([11, 12, 13], 12)],
                  33: [([1, 2, 3], 2), ([9, 10, 11], 10), ([19, 20, 21], 18), ([29, 30, 31], 31)],
                  42: [([1, 2, 3], 2), ([11, 12, 13], 12), ([21, 22, 23], 22), ([31, 32, 33], 32), ([39, 40, 41], 40)]}
        if n in groups:
            atom_types = np.arange(n)
            for idx, v in groups[n]:
                atom_types[idx] = v
            return torch.nn.functional.one_hot(torch.tensor(atom_types))
        if n in (13, 55):
            return torch.zeros(n, 1)
        raise NotImplementedError(
            f"egnn_aldp.EGNN_dynamics: the node features of {n} particles come from a topology (:80-129); pass "
            "h_initial=[n_particles, n_features] instead")

    def forward(self, t, xs, beta=None):
        # the reference's forward takes beta positionally and ignores it unless condition_temperature (:131-146)
        return super().forward(t, xs, beta if self.condition_temperature else None)
