"""Level-scheduled evaluation of NAND netlists: every gate of a dependency level goes into one
batched call (reference: each gate is one HomNAND::nand call, src/hom_nand.hpp:74-81; the
reference's own chained network is test/main.cpp:192-205).

A netlist is a list of gates (out_wire, in_a, in_b) over integer wire ids.  Wires that no gate
drives are inputs.  `levelize` assigns every gate the earliest level at which both operands
exist; `evaluate` walks the levels, gathering operand rows from a [n_wires, n+1] wire array,
calling a batched NAND, and scattering the results back.
"""
import numpy as np

MU = 1 << 29


def trivial_bit(bit, n1):
    """Noiseless ciphertext of a constant: b = +-1/8, mask = 0 (decrypts under any key)."""
    ct = np.zeros(n1, dtype=np.uint32)
    ct[0] = MU if bit else (1 << 32) - MU
    return ct


def levelize(gates, n_wires):
    """Returns a list of levels, each a list of gate indices.  Raises on a combinational loop
    or a wire driven twice."""
    driver = {}
    for g, (o, _, _) in enumerate(gates):
        if o in driver:
            raise ValueError(f"wire {o} is driven by gates {driver[o]} and {g}")
        if not 0 <= o < n_wires:
            raise ValueError(f"gate {g} drives wire {o} outside [0, {n_wires})")
        driver[o] = g
    level_of_wire = {}
    level_of_gate = [None] * len(gates)

    def wire_level(w, stack):
        if w not in driver:
            return 0
        if w in level_of_wire:
            return level_of_wire[w]
        g = driver[w]
        if g in stack:
            raise ValueError("combinational loop through wire %d" % w)
        stack.add(g)
        _, a, b = gates[g]
        lv = 1 + max(wire_level(a, stack), wire_level(b, stack))
        stack.discard(g)
        level_of_wire[w] = lv
        level_of_gate[g] = lv
        return lv

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 4 * len(gates) + 100))
    try:
        for g, (o, _, _) in enumerate(gates):
            wire_level(o, set())
    finally:
        sys.setrecursionlimit(old)
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


def evaluate(gates, wires, nand_batch, levels=None):
    """Evaluates the netlist in place on `wires` ([n_wires, n+1] numpy uint32, or a torch tensor
    when nand_batch works on tensors).  nand_batch(a_rows, b_rows) -> out_rows.  With tensors,
    nand_batch must enqueue on torch's current stream (Context.nand_device(..., stream=
    torch.cuda.current_stream())) so that it is ordered with the gathers and scatters here.
    Returns the number of levels executed."""
    if levels is None:
        levels = levelize(gates, wires.shape[0])
    is_torch = not isinstance(wires, np.ndarray)
    for lv in levels:
        o = [gates[g][0] for g in lv]
        ia = [gates[g][1] for g in lv]
        ib = [gates[g][2] for g in lv]
        if is_torch:
            import torch

            dev = wires.device
            ia_t, ib_t, o_t = (torch.tensor(v, dtype=torch.long, device=dev) for v in (ia, ib, o))
            out = nand_batch(wires.index_select(0, ia_t).contiguous(), wires.index_select(0, ib_t).contiguous())
            wires.index_copy_(0, o_t, out)
        else:
            wires[o] = nand_batch(np.ascontiguousarray(wires[ia]), np.ascontiguousarray(wires[ib]))
    return len(levels)
