"""``spconv.modules`` counterpart: SparseModule, SparseSequential, ToDense (spconv 1.2.1,
SURVEY.md A.1; reference call sites src/models/SPConvBlocks.py:81-82,515-516,822)."""
from collections import OrderedDict

import torch
from torch import nn


class SparseModule(nn.Module):
    """place holder, all module subclass from this will take sptensor in SparseSequential."""
    pass


def _is_sparse_tensor(x):
    from .tensor import SparseConvTensor
    return isinstance(x, SparseConvTensor)


class _SideJoin(object):
    """The end of a side-stream branch, waited for at most once per consuming stream."""

    def __init__(self, side):
        self.event = torch.cuda.Event()
        self.event.record(side)
        self.waited = set()

    def wait(self, stream=None):
        stream = stream if stream is not None else torch.cuda.current_stream()
        key = stream.cuda_stream
        if key not in self.waited:
            self.waited.add(key)
            stream.wait_event(self.event)


class SparseSequential(SparseModule):
    """Sequential container: spconv modules receive the SparseConvTensor, plain ``nn.Module``s are
    applied to ``.features`` (skipped when there are no active rows); after ``ToDense`` the value
    is a dense tensor and later modules receive it directly."""

    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for i in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    @staticmethod
    def _dense_follows(mods, i):
        """Is the next sparse module after mods[i] a ToDense?  (The conv at i then asks its rulebook build for the
        cell -> row map that dense() of its output uses; csrc/evconv.hip writes it only on request.)"""
        for m in mods[i + 1:]:
            if isinstance(m, SparseModule):
                return isinstance(m, ToDense)
        return False

    @staticmethod
    def _prefetch_rulebooks(mods, x, owner=None):
        """Build every layer's rulebook on a side stream now (device-count mode: nothing synchronises with the
        host).  Rulebooks depend on indices only, so the strided layers' builds run while the first layers
        compute; each conv waits for its own rulebook's event.  ``owner``: a submanifold layer about to run on the
        calling stream -- it and the layers sharing its indice_key keep building / finding their rulebook there, the
        branch forks BEFORE it (its output has the input's rows, so nothing on the branch depends on it)."""
        from . import ops
        from .conv import SparseConvolution
        main = torch.cuda.current_stream()
        side = ops.side_stream(x.features.device)
        side.wait_stream(main)
        plan = {}
        built = []
        keyed = {k: v.rulebook for k, v in x.indice_dict.items() if hasattr(v, "rulebook")}      # already built
        indices, spatial, n_dev = x.indices, x.spatial_shape, x.n_valid
        events = getattr(x, "events", None)
        with torch.cuda.stream(side):
            for at, m in enumerate(mods):
                if isinstance(m, SparseConvolution):
                    if m.conv1x1:
                        continue
                    if m.inverse or m.transposed:
                        break                       # geometry comes from a coupled layer: leave the rest to the layers
                    if owner is not None and (m is owner or (m.subm and m.indice_key is not None
                                                             and m.indice_key == owner.indice_key)):
                        continue
                    if m.indice_key is not None and m.indice_key in keyed:
                        rb = keyed[m.indice_key]
                    else:
                        rb = ops.build_rulebook(indices, x.batch_size, spatial, m.kernel_size, m.stride, m.padding,
                                                m.dilation, m.subm, known_unique=True, n_dev=n_dev,
                                                out_capacity=getattr(m, "out_capacity", None), events=events,
                                                flags=m._sticky_flags(),
                                                want_cell_map=SparseSequential._dense_follows(mods, at))
                        built.append(rb)
                        if ops.JOIN_PER_BUILD:
                            rb.ready = _SideJoin(side)      # its own edge: the layer waits for THIS build only
                        if m.indice_key is not None:
                            keyed[m.indice_key] = rb
                    plan[id(m)] = rb
                    if not m.subm:
                        indices, spatial, n_dev = rb.out_indices, rb.out_spatial_shape, rb.m_dev
                        events = getattr(rb, "events_out", None)
                elif isinstance(m, SparseModule):
                    break                           # ToDense or an unknown sparse module ends the sparse stack
            if built and not ops.JOIN_PER_BUILD:
                # ONE join for the whole branch: the first layer that needs any of these rulebooks waits for all of them
                # (every further cross-stream edge costs a captured step 5 - 10 us)
                join = _SideJoin(side)
                for rb in built:
                    rb.ready = join
        x.prefetched = plan

    def forward(self, input):
        return self.run(input, list(self._modules.values()))

    def run(self, input, mods, stop_before_dense=False):
        """``forward`` over an explicit module list.  ``stop_before_dense``: a TRAILING ToDense is not applied -- the
        SparseConvTensor in front of it is returned (psd/net.py hands it to the sparse head, functional.sparse_head),
        while the layers still see that a ToDense follows (the last conv's build leaves its cell -> row map)."""
        from . import functional as Fsp
        from . import ops
        want_prefetch = (ops.PREFETCH_RULEBOOKS and _is_sparse_tensor(input) and input.n_valid is not None
                         and getattr(input, "prefetched", None) is None and input.features.is_cuda)
        i = 0
        while i < len(mods):
            module = mods[i]
            if stop_before_dense and i == len(mods) - 1 and isinstance(module, ToDense) and _is_sparse_tensor(input):
                break
            if isinstance(module, SparseModule):
                if _is_sparse_tensor(input):
                    input.dense_follows = self._dense_follows(mods, i)
                if (want_prefetch and ops.PREFETCH_BEFORE_FIRST and getattr(input, "events", None) is not None
                        and getattr(module, "subm", False) and not getattr(module, "conv1x1", False)
                        and input.find_indice_pair(module.indice_key) is None):
                    # the event offsets came with the batch: the strided layers' builds need nothing the first layer
                    # makes, so the branch forks before its build and its conv rather than after them
                    want_prefetch = False
                    self._prefetch_rulebooks(mods[i:], input, owner=module)
                input = module(input)
                if want_prefetch and _is_sparse_tensor(input):
                    # the first layer has built its own rulebook and launched its conv on this stream; the
                    # remaining layers' rulebooks now build on the side stream beside what follows
                    want_prefetch = False
                    self._prefetch_rulebooks(mods[i + 1:], input)
            elif _is_sparse_tensor(input):
                if input.indices.shape[0] != 0:
                    if isinstance(module, nn.BatchNorm1d) and Fsp.can_fuse_batch_norm(module, input.features):
                        # BatchNorm1d [+ ReLU] over the active rows: one fused pair of HIP launches
                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        input.features = Fsp.batch_norm_relu(input.features, module, relu, input.n_valid)
                        if relu:
                            i += 1
                    else:
                        if input.n_valid is not None and type(module) is not nn.ReLU:      # ReLU is row-local
                            raise RuntimeError("device-count (padded) SparseConvTensors only support the fused "
                                               "BatchNorm1d/ReLU pair between sparse layers, got %s" % type(module).__name__)
                        input.features = module(input.features)
            else:
                input = module(input)
            i += 1
        return input


class ToDense(SparseModule):
    """convert SparseConvTensor to NCHW dense tensor."""

    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):
    """remove pre-allocated grid buffer."""

    def forward(self, x):
        x.grid = None
        return x
