"""Variable store + name scopes: the host-side stand-in for tf.get_variable / tf.variable_scope.

Variables are created up-front in the reference's creation order so that the
flat parameter slabs can be laid out once; the L1 op shims then look them up by their TF name
(``Discriminator/D.Block.1.Conv1/Filters`` ...), which is also the name used in checkpoints.
"""
import contextlib

from . import ops as O

_scope_stack = []


@contextlib.contextmanager
def variable_scope(name):
    _scope_stack.append(name)
    try:
        yield "/".join(_scope_stack)
    finally:
        _scope_stack.pop()


def scoped(name):
    return "/".join(_scope_stack + [name])


class Graph:
    """What the L1 shims need to resolve a variable during one step: the parameter groups, the
    non-trainable state (SN u, BN moving stats) and the spectral-norm results prefetched for this step."""

    current = None

    def __init__(self, ctx, groups, state):
        self.ctx = ctx
        self.groups = groups            # list of ParamGroup
        self.state = state              # name -> persistent DT
        self.trainable = set()          # group indices whose params require grad this step
        self._params = {}
        self.sn = {}                    # param name -> (Weight, update flag) prefetched this step
        self.plain = {}                 # param name -> Weight (no SN), per step
        self.persist = {}               # param name -> {layout key: DT}: prepared filters that outlive the step
        self.index = {}
        for gi, g in enumerate(groups):
            for n in g.names:
                self.index[n] = gi

    def begin_step(self, trainable_groups):
        self.trainable = set(trainable_groups)
        self._params = {}
        self.sn = {}
        self.plain = {}
        # per-step riders' outputs never outlive the step that produced them (the critic step's pooled images, the head's
        # label embeddings): a later forward-only pass must not pick up a previous step's tensors
        self.image_pool = None
        self.head_E = None
        self.trunk_frag = None
        # overlapped data-parallel schedule (RCGAN_DP_OVERLAP=1): closures the model functions record on the tape where the LAST layers'
        # gradients are complete in the backward pass (Discriminator: in front of D.Block.3; Generator: in front of G.Block.2) -- they
        # finish that bucket and hand it to the all-reduce while the earlier layers' backward still runs (CifarRCGAN._dp_early)
        self.early_d = None
        self.early_g = None
        Graph.current = self

    def param(self, name):
        if name not in self._params:
            if name not in self.index:
                raise KeyError("Variable %s does not exist (variables are created up-front)" % name)
            gi = self.index[name]
            p = self.groups[gi].param(name)
            p.req = gi in self.trainable
            self._params[name] = p
        return self._params[name]

    def has(self, name):
        return name in self.index

    def prefetch_sn(self, entries):
        """entries: list of (param name, u-state name, update).  One batched launch for all of them."""
        ents = [(self.param(pn), self.state[un], upd) for pn, un, upd in entries]
        ws = O.spectral_norm_batch(self.ctx, ents)
        for (pn, _, upd), w in zip(entries, ws):
            self.sn[pn] = (w, upd)

    def sn_weight(self, pname, uname, update):
        if pname not in self.sn:
            self.prefetch_sn([(pname, uname, update)])
        w, upd = self.sn[pname]
        if bool(upd) != bool(update):
            raise ValueError("spectral norm of %s was prefetched with update=%s but used with update=%s" % (pname, upd, update))
        return w

    def prepare_convs(self, names, dtype, embed=None, inputs=None, frags=None):
        """Batch-prepare the conv filters named in ``names`` ([(param name, k, stride, input hw)]): one launch.
        embed / inputs / frags: see ops.prepare_batch; returns whether the riders rode along (and, with frags, which Weights' fragment
        copies the launch wrote)."""
        ws = []
        for pn, k, stride, hw, *rest in names:
            w = self.sn[pn][0] if pn in self.sn else self.weight(pn)
            ws.append((w, k, stride, hw, *rest))
        return O.prepare_batch(self.ctx, ws, dtype, embed=embed, inputs=inputs, frags=frags)

    def refresh_persistent(self, names, dtype):
        """(Re)prepare the un-normalised filters in ``names`` into buffers that survive the step: called after their
        optimiser step (or any other write to the parameters), NOT once per step."""
        ws = [(O.Weight(self.ctx, self.groups[self.index[pn]].param(pn), None), k, stride, hw, *rest) for pn, k, stride, hw, *rest in names]
        O.prepare_batch(self.ctx, ws, dtype, persistent=self.persist)

    def weight(self, pname):
        if pname not in self.plain:
            w = O.Weight(self.ctx, self.param(pname), None)
            w._prepared.update(self.persist.get(pname, {}))
            self.plain[pname] = w
        return self.plain[pname]
