"""Level-scheduled evaluation of gate netlists: every gate of a dependency level goes into one
batched call (reference: each gate is one HomNAND::nand call, src/hom_nand.hpp:74-81; the
reference's own chained network is test/main.cpp:192-205).

A netlist is a list of gates over integer wire ids:
    (out, a, b)              NAND, the reference's only gate
    (out, a, b, op)          op in binding.GATES ("and", "or", "xor", ...): the same bootstrap
                             behind another linear combination of the operands
    (out, a, None, "not")    negation, no bootstrap
    (out, sel, x1, "mux", x0)   sel ? x1 : x0, two blind rotations and one key switch
Wires that no gate drives are inputs.  `levelize` assigns every gate the earliest level at which
all its operands exist; `evaluate` walks the levels, gathering operand rows from a
[n_wires, n+1] wire array, calling the backend once per kind of gate present in the level
(all two-input gates of a level share one call), and scattering the results back.
"""
import numpy as np

from .binding import GATES

MU = 1 << 29


def trivial_bit(bit, n1):
    """Noiseless ciphertext of a constant: b = +-1/8, mask = 0 (decrypts under any key)."""
    ct = np.zeros(n1, dtype=np.uint32)
    ct[0] = MU if bit else (1 << 32) - MU
    return ct


def normalize(gate):
    """(out, op, inputs) of a netlist entry."""
    if len(gate) == 3:
        return gate[0], "nand", (gate[1], gate[2])
    op = gate[3]
    if op == "not":
        return gate[0], op, (gate[1],)
    if op == "mux":
        if len(gate) != 5:
            raise ValueError("mux gates are (out, sel, x1, 'mux', x0)")
        return gate[0], op, (gate[1], gate[2], gate[4])
    if op not in GATES:
        raise ValueError(f"unknown gate {op!r}")
    return gate[0], op, (gate[1], gate[2])


def levelize(gates, n_wires):
    """Returns a list of levels, each a list of gate indices.  Raises on a combinational loop
    or a wire driven twice."""
    norm = [normalize(g) for g in gates]
    driver = {}
    for g, (o, _, _) in enumerate(norm):
        if o in driver:
            raise ValueError(f"wire {o} is driven by gates {driver[o]} and {g}")
        if not 0 <= o < n_wires:
            raise ValueError(f"gate {g} drives wire {o} outside [0, {n_wires})")
        driver[o] = g
    level_of_gate = [None] * len(gates)
    # iterative depth-first search: state 0 = enter, 1 = operands done
    for root in range(len(gates)):
        if level_of_gate[root] is not None:
            continue
        stack = [(root, 0)]
        on_path = set()
        while stack:
            g, state = stack.pop()
            if state == 0:
                if level_of_gate[g] is not None:
                    continue
                if g in on_path:
                    raise ValueError("combinational loop through wire %d" % norm[g][0])
                on_path.add(g)
                stack.append((g, 1))
                for w in norm[g][2]:
                    d = driver.get(w)
                    if d is not None and level_of_gate[d] is None:
                        if d in on_path:
                            raise ValueError("combinational loop through wire %d" % w)
                        stack.append((d, 0))
            else:
                on_path.discard(g)
                level_of_gate[g] = 1 + max((level_of_gate[driver[w]] if w in driver else 0) for w in norm[g][2])
    depth = max(level_of_gate, default=0)
    levels = [[] for _ in range(depth)]
    for g, lv in enumerate(level_of_gate):
        levels[lv - 1].append(g)
    return levels


def ripple_carry_adder(bits):
    """NAND-only ripple-carry adder, 9 gates per full adder:
        n1 = a^b' ... : n1 = a NAND b, n2 = a NAND n1, n3 = b NAND n1, x = n2 NAND n3 (a xor b),
        n4 = x NAND cin, n5 = x NAND n4, n6 = cin NAND n4, sum = n5 NAND n6, cout = n1 NAND n4.
    Wires: a[0..bits), b[0..bits), zero (carry-in of bit 0), then the gate outputs.
    Returns (gates, n_wires, a_wires, b_wires, zero_wire, sum_wires, carry_out_wire)."""
    a = list(range(bits))
    b = list(range(bits, 2 * bits))
    zero = 2 * bits
    nxt = 2 * bits + 1
    gates, sums = [], []
    cin = zero
    for i in range(bits):
        n1, n2, n3, x, n4, n5, n6, s, cout = range(nxt, nxt + 9)
        nxt += 9
        gates += [(n1, a[i], b[i]), (n2, a[i], n1), (n3, b[i], n1), (x, n2, n3), (n4, x, cin), (n5, x, n4),
                  (n6, cin, n4), (s, n5, n6), (cout, n1, n4)]
        sums.append(s)
        cin = cout
    return gates, nxt, a, b, zero, sums, cin


def mux_adder(bits):
    """Ripple-carry adder with one gate per bit on the carry chain: x = a xor b,
    cout = x ? cin : a, sum = x xor cin.  3 gates (4 blind rotations) per bit and bits + 2 levels,
    against 9 gates and 2 bits + 4 levels for the NAND-only form.  Same return layout as
    ripple_carry_adder."""
    a = list(range(bits))
    b = list(range(bits, 2 * bits))
    zero = 2 * bits
    nxt = 2 * bits + 1
    gates, sums = [], []
    cin = zero
    for i in range(bits):
        x, s, cout = range(nxt, nxt + 3)
        nxt += 3
        gates += [(x, a[i], b[i], "xor"), (cout, x, cin, "mux", a[i]), (s, x, cin, "xor")]
        sums.append(s)
        cin = cout
    return gates, nxt, a, b, zero, sums, cin


def prune(gates, outputs):
    """The gates that some wire of `outputs` depends on, in their original order."""
    norm = [normalize(g) for g in gates]
    driver = {o: g for g, (o, _, _) in enumerate(norm)}
    keep, stack = set(), [driver[w] for w in outputs if w in driver]
    while stack:
        g = stack.pop()
        if g in keep:
            continue
        keep.add(g)
        stack.extend(driver[w] for w in norm[g][2] if w in driver)
    return [gates[g] for g in sorted(keep)]


def prefix_adder(bits):
    """Parallel-prefix (Kogge-Stone) adder: the same sum as the ripple-carry netlists in
    log2(bits) + 2 levels instead of bits + 2, at the price of more gates per level -- which a
    batched bootstrap evaluates in the time of one.
        level 1:            g_i = a_i and b_i,  p_i = a_i xor b_i
        distance d = 1, 2, 4, ...:   G_i = P_i ? G_(i-d) : G_i,   P_i = P_i and P_(i-d)     (i >= d)
        last level:         sum_i = p_i xor G_(i-1)   (sum_0 = p_0),   carry out = G_(bits-1)
    A group's generate and propagate signals are never both 1, so "G or (P and G')" is the MUX
    above: one gate level per distance.  Gates that no output depends on are dropped.  Same return
    layout as ripple_carry_adder (the `zero` wire is unused)."""
    a = list(range(bits))
    b = list(range(bits, 2 * bits))
    zero = 2 * bits
    nxt = 2 * bits + 1
    gates = []

    def new():
        nonlocal nxt
        nxt += 1
        return nxt - 1

    g, prop = [], []
    for i in range(bits):
        gi, pi = new(), new()
        gates += [(gi, a[i], b[i], "and"), (pi, a[i], b[i], "xor")]
        g.append(gi)
        prop.append(pi)
    G, P = list(g), list(prop)
    d = 1
    while d < bits:
        G2, P2 = list(G), list(P)
        for i in range(d, bits):
            G2[i], P2[i] = new(), new()
            gates += [(G2[i], P[i], G[i - d], "mux", G[i]), (P2[i], P[i], P[i - d], "and")]
        G, P = G2, P2
        d *= 2
    sums = [prop[0]]
    for i in range(1, bits):
        s = new()
        gates.append((s, prop[i], G[i - 1], "xor"))
        sums.append(s)
    cout = G[bits - 1]
    return prune(gates, sums + [cout]), nxt, a, b, zero, sums, cout


class HostBackend:
    """numpy rows through the host-pointer entry points: one call per gate type in a level."""

    def __init__(self, ctx):
        self.ctx = ctx

    def gates(self, ops, a, b):
        out = np.empty_like(a)
        ops = np.asarray(ops)
        for op in np.unique(ops):
            sel = np.nonzero(ops == op)[0]
            out[sel] = self.ctx.gate(int(op), np.ascontiguousarray(a[sel]), np.ascontiguousarray(b[sel]))
        return out

    def not_(self, a):
        return self.ctx.not_(a)

    def mux(self, sel, x1, x0):
        return self.ctx.mux(sel, x1, x0)


class DeviceBackend:
    """torch GPU rows through the device-pointer entry points, enqueued on torch's current stream
    so that they are ordered with the gathers and scatters of `evaluate`; all two-input gates of
    a level go into one call with per-gate codes."""

    def __init__(self, ctx):
        import torch

        self.ctx = ctx
        self.torch = torch

    def _stream(self):
        return self.torch.cuda.current_stream()

    def _codes(self, ops, device):
        if hasattr(ops, "data_ptr"):
            return ops
        return self.torch.tensor(np.asarray(ops, dtype=np.uint8), device=device)

    def gates(self, ops, a, b):
        return self.ctx.gate_device(self._codes(ops, a.device), a, b, self.torch.empty_like(a), stream=self._stream())

    def not_(self, a):
        return self.ctx.not_device(a, self.torch.empty_like(a), stream=self._stream())

    def mux(self, sel, x1, x0):
        return self.ctx.mux_device(sel, x1, x0, self.torch.empty_like(sel), stream=self._stream())

    def level(self, ops, a, b, x0, n_mux):
        """Two-input gates (codes `ops`) followed by n_mux MUX gates, one launch."""
        ops_t = self._codes(ops, a.device) if len(ops) else None
        return self.ctx.level_device(ops_t, a, b, x0, self.torch.empty_like(a), n_mux, stream=self._stream())


class _NandOnly:
    """Adapter for a plain nand_batch(a, b) callable."""

    def __init__(self, fn):
        self.fn = fn

    def gates(self, ops, a, b):
        if bool((ops != 0).any()):
            raise ValueError("this backend evaluates NAND gates only")
        return self.fn(a, b)


class Plan:
    """A levelised netlist with its per-level operand and result index arrays prepared once, so
    that evaluating the circuit again (other inputs, same netlist) builds nothing and, with GPU
    tensors, copies no index array to the device."""

    def __init__(self, gates, n_wires, levels=None):
        self.n_wires = n_wires
        norm = [normalize(g) for g in gates]
        self.levels = levelize(gates, n_wires) if levels is None else levels
        self.steps = []
        for lv in self.levels:
            two = [g for g in lv if norm[g][1] in GATES]
            nots = [g for g in lv if norm[g][1] == "not"]
            muxes = [g for g in lv if norm[g][1] == "mux"]
            idx = lambda gs, k: np.array([norm[g][2][k] for g in gs], dtype=np.int64)  # noqa: E731
            outs = lambda gs: np.array([norm[g][0] for g in gs], dtype=np.int64)  # noqa: E731
            self.steps.append({
                "ops": np.array([GATES[norm[g][1]] for g in two], dtype=np.uint8),
                "two": (idx(two, 0), idx(two, 1), outs(two)),
                "not": (idx(nots, 0), outs(nots)),
                "mux": (idx(muxes, 0), idx(muxes, 1), idx(muxes, 2), outs(muxes)),
                # two-input gates followed by MUXes, for backends that take a whole level at once
                "both": (np.concatenate([idx(two, 0), idx(muxes, 0)]), np.concatenate([idx(two, 1), idx(muxes, 1)]),
                         np.concatenate([outs(two), outs(muxes)])),
            })
        self._device_steps = {}

    def on(self, device):
        """The same arrays as torch tensors on `device` (cached)."""
        import torch

        key = str(device)
        if key not in self._device_steps:
            conv = lambda v: torch.from_numpy(v).to(device)  # noqa: E731
            self._device_steps[key] = [{k: (conv(v) if isinstance(v, np.ndarray) else tuple(conv(x) for x in v))
                                        for k, v in st.items()} for st in self.steps]
        return self._device_steps[key]


def evaluate(gates, wires, backend, levels=None, plan=None):
    """Evaluates the netlist in place on `wires` ([n_wires, n+1] numpy uint32, or a torch GPU
    tensor with a DeviceBackend).  `backend`: a HostBackend / DeviceBackend (or any object with
    gates(ops, a, b), not_(a), mux(sel, x1, x0) and optionally level(ops, a, b, x0, n_mux)), or a
    callable nand_batch(a_rows, b_rows) -> out_rows for NAND-only netlists.  `plan`: a Plan of
    the same netlist, to reuse its index arrays across calls.  Returns the number of levels."""
    if callable(backend) and not hasattr(backend, "gates"):
        backend = _NandOnly(backend)
    if plan is None:
        plan = Plan(gates, wires.shape[0], levels)
    is_torch = not isinstance(wires, np.ndarray)
    if is_torch:
        steps = plan.on(wires.device)

        def rows(idx):
            return wires.index_select(0, idx)

        def store(idx, val):
            wires.index_copy_(0, idx, val)
    else:
        steps = plan.steps

        def rows(idx):
            return np.ascontiguousarray(wires[idx])

        def store(idx, val):
            wires[idx] = val

    for st in steps:
        n_two, n_not, n_mux = len(st["ops"]), len(st["not"][0]), len(st["mux"][0])
        # all operands are read before any result of the level is stored
        results = []
        if n_two and n_mux and hasattr(backend, "level"):
            # both kinds in one call: the level then costs one blind-rotation latency
            ia, ib, o = st["both"]
            results.append((o, backend.level(st["ops"], rows(ia), rows(ib), rows(st["mux"][2]), n_mux)))
            n_two = n_mux = 0
        if n_two:
            ia, ib, o = st["two"]
            results.append((o, backend.gates(st["ops"], rows(ia), rows(ib))))
        if n_not:
            ia, o = st["not"]
            results.append((o, backend.not_(rows(ia))))
        if n_mux:
            isel, i1, i0, o = st["mux"]
            results.append((o, backend.mux(rows(isel), rows(i1), rows(i0))))
        for idx, val in results:
            store(idx, val)
    return len(plan.levels)
