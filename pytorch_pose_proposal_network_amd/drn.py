"""Mirror of the reference's DRN-D factories (drn.py:345-398): each returns a spec the HIP
``PoseProposalNet`` lowers into its conv program.  ``pretrained=True`` is refused: the reference fetches
weights over the network (drn.py:7-18), which is out of scope and offline here."""
from __future__ import annotations

from .model import DRNSpec


def _factory(name):
    def make(pretrained: bool = False, **kwargs):
        if pretrained:
            raise RuntimeError(f"{name}(pretrained=True) needs network access; load a checkpoint with load_state_dict")
        return DRNSpec(name)
    make.__name__ = name
    return make


drn_d_22 = _factory("drn_d_22")
drn_d_24 = _factory("drn_d_24")
drn_d_38 = _factory("drn_d_38")
drn_d_40 = _factory("drn_d_40")
drn_d_54 = _factory("drn_d_54")
drn_d_56 = _factory("drn_d_56")
drn_d_105 = _factory("drn_d_105")
drn_d_107 = _factory("drn_d_107")
