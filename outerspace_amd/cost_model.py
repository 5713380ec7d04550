"""OuterSPACE analytical cost model -- the closed-form "predicted cycles / DRAM bytes" report.

Restates `simulateOuterSPACEAnalytical` of the reference (simulator/SimOuterSPACE.cpp:176-238) and the task
shapes its `TaskProvider` builds (:44-132), vectorised over torch tensors so that it runs on the GPU next to
the measured product (O(nnzA) work, no partial products are formed).

The model, as the reference computes it (config = OuterSPACEConfig, :17-25: NUM_PE 256, BLOCK_SIZE 64,
DRAM_BANDWIDTH = size_t(16 * 8 / 1.5) = 85 bytes per cycle):

* one MULTIPLY task per k with nnz(A[:,k]) > 0 and nnz(B[k,:]) > 0 (:74-98), in k order;
  workload = nnz(A[:,k]) * nnz(B[k,:]);
  dram = align64(workload * S) + align64(nnz(A[:,k]) * S) + align64(nnz(B[k,:]) * S), S = sizeof(CSRElement) (:176-181);
* one MERGE task per output row i < max row index of A + 1 (:49-53, :99-132); its inputs are the chunks
  (one per non-zero A[i,k] with a non-empty B row), `ways` of them with U_i entries in total;
  workload = U_i * ways (:183-188); dram = sum align64(chunk * S) + align64(out * S) (:190-194);
* cycles of a task = max(workload, dram * NUM_PE / DRAM_BANDWIDTH) in integer arithmetic (:198-202);
* tasks are dealt round-robin to the PEs (task t -> PE t % NUM_PE, :155-161); a phase costs the busiest PE's sum;
  the total is multiply + merge (:204-238).

Reference quirk kept and reported separately: `TaskProvider` labels partial products with their POSITION in the
B row instead of their column (:88-90) and its merge loop pushes on EQUAL neighbours (:119-126), so the
`out` it prices for row i is 1 + U_i - max_k nnz(B[k,:]) rather than nnz(C[i,:]).  `output="reference"` reproduces
that; `output="nnz"` prices the real merged row lengths (pass `c_rowptr`).

Parity: UNPINNED against a reference build -- SimOuterSPACE.cpp needs ramulator's Memory.h, SimCycle.cpp and
parameters.cpp, none of which are in the reference tree, so it cannot be compiled here; the
restatement is checked against a literal pure-Python replay of the same lines (test infrastructure, kept with the
other checkers) and hand-computed cases (tests/test_cost_model.py)."""
from __future__ import annotations

import torch

NUM_PE = 256            # SimOuterSPACE.cpp:18
BLOCK_SIZE = 64         # :20
DRAM_BANDWIDTH = int(16 * 8 / 1.5)  # :24, a size_t: 85


def _align(x: torch.Tensor, a: int) -> torch.Tensor:
    """common.h:59 alignTo: round up to a multiple of `a`."""
    return (x + (a - 1)) // a * a


def _cycles(workload: torch.Tensor, dram: torch.Tensor) -> torch.Tensor:
    return torch.maximum(workload, dram * NUM_PE // DRAM_BANDWIDTH)   # :198-202


def _busiest_pe(cycles: torch.Tensor) -> int:
    """Round-robin dispatch (:155-161): task t runs on PE t % NUM_PE; the phase takes the busiest PE's total."""
    if cycles.numel() == 0:
        return 0
    pe = torch.arange(cycles.numel(), device=cycles.device) % NUM_PE
    return int(torch.zeros(NUM_PE, dtype=torch.int64, device=cycles.device).index_add_(0, pe, cycles).max())


def analytical(a_colptr, a_rowidx, b_rowptr, value_size: int = 4, c_rowptr=None, output: str = "reference") -> dict:
    """Predicted cycles and DRAM bytes of C = A * B for CSC(A) (colptr, rowidx) and CSR(B) (rowptr).

    value_size: sizeof(value_t) -- 4 in the reference (CSRElement = 8 B), 8 for the f64 build (12 B, packed).
    Returns {"cycles_multiply", "cycles_merge", "cycles_total", "dram_bytes_multiply", "dram_bytes_merge",
             "multiply_tasks", "merge_tasks", "workload_multiply", "workload_merge"} (python ints)."""
    if output not in ("reference", "nnz"):
        raise ValueError("output must be 'reference' or 'nnz'")
    if output == "nnz" and c_rowptr is None:
        raise ValueError("output='nnz' needs c_rowptr")
    a_colptr = torch.as_tensor(a_colptr).to(torch.int64)
    dev = a_colptr.device
    a_rowidx = torch.as_tensor(a_rowidx).to(device=dev, dtype=torch.int64)
    b_rowptr = torch.as_tensor(b_rowptr).to(device=dev, dtype=torch.int64)
    if a_colptr.numel() != b_rowptr.numel():
        raise ValueError("A (CSC) and B (CSR) must share the inner dimension")  # the assert at :46
    S = 4 + int(value_size)
    acnt = a_colptr[1:] - a_colptr[:-1]
    bcnt = b_rowptr[1:] - b_rowptr[:-1]

    # ---- multiply tasks (:74-98, :176-181) ----
    active = (acnt > 0) & (bcnt > 0)
    wa, wb = acnt[active], bcnt[active]
    work_mul = wa * wb
    dram_mul = _align(work_mul * S, BLOCK_SIZE) + _align(wa * S, BLOCK_SIZE) + _align(wb * S, BLOCK_SIZE)
    cyc_mul = _busiest_pe(_cycles(work_mul, dram_mul))

    # ---- merge tasks (:99-132, :183-194): one per row below max row index + 1, empty ones included ----
    nnz_a = int(a_rowidx.numel())
    if nnz_a == 0:
        nrows = 1  # maxRowId stays 0 (:49-53)
        ways = u = dram_in = maxb = torch.zeros(1, dtype=torch.int64, device=dev)
    else:
        nrows = int(a_rowidx.max()) + 1
        k_of = torch.repeat_interleave(torch.arange(acnt.numel(), device=dev), acnt)   # column of every A entry
        chunk = bcnt[k_of]
        hit = chunk > 0                                                                 # inactive k produce no chunk
        rows, chunk = a_rowidx[hit], chunk[hit]
        z = lambda: torch.zeros(nrows, dtype=torch.int64, device=dev)
        ways = z().index_add_(0, rows, torch.ones_like(chunk))
        u = z().index_add_(0, rows, chunk)
        dram_in = z().index_add_(0, rows, _align(chunk * S, BLOCK_SIZE))
        maxb = z().scatter_reduce_(0, rows, chunk, reduce="amax", include_self=True)
    if output == "reference":
        out = torch.where(u > 0, 1 + u - maxb, torch.zeros_like(u))   # the quirk described above
    else:
        c_rowptr = torch.as_tensor(c_rowptr).to(device=dev, dtype=torch.int64)
        out = (c_rowptr[1:] - c_rowptr[:-1])[:nrows]
        if out.numel() < nrows:
            out = torch.cat([out, torch.zeros(nrows - out.numel(), dtype=torch.int64, device=dev)])
    work_mer = u * ways
    dram_mer = dram_in + _align(out * S, BLOCK_SIZE)
    cyc_mer = _busiest_pe(_cycles(work_mer, dram_mer))
    return {
        "cycles_multiply": cyc_mul, "cycles_merge": cyc_mer, "cycles_total": cyc_mul + cyc_mer,
        "dram_bytes_multiply": int(dram_mul.sum()), "dram_bytes_merge": int(dram_mer.sum()),
        "multiply_tasks": int(active.sum()), "merge_tasks": nrows,
        "workload_multiply": int(work_mul.sum()), "workload_merge": int(work_mer.sum()),
        "config": {"NUM_PE": NUM_PE, "BLOCK_SIZE": BLOCK_SIZE, "DRAM_BANDWIDTH": DRAM_BANDWIDTH, "sizeof_CSRElement": S,
                   "output": output},
    }
