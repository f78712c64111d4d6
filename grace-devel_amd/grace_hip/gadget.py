"""Gadget-2 (format 1) snapshot I/O for the SPH fields GRACE uses: gas positions and
smoothing lengths.  Block layout as read by the reference's tests/helper/read_gadget.cuh:
header of 256 B (npart[6] int32, mass[6] float64, padding) and POS, VEL, ID, [MASS], U, RHO,
HSML blocks, every block framed by 4-byte markers.  Host-side file I/O only (numpy)."""
import numpy as np


def write_gadget(fname, pos, hsml, masses_in_header=True):
    """Writes a gas-only snapshot: pos [N,3] float32, hsml [N] float32."""
    pos = np.ascontiguousarray(pos, np.float32); hsml = np.ascontiguousarray(hsml, np.float32)
    n = len(pos)
    npart = np.array([n, 0, 0, 0, 0, 0], np.int32)
    mass = np.array([1.0 if masses_in_header else 0.0, 0, 0, 0, 0, 0], np.float64)

    def block(f, payload):
        nbytes = np.array([len(payload)], np.int32).tobytes()
        f.write(nbytes); f.write(payload); f.write(nbytes)

    with open(fname, "wb") as f:
        header = npart.tobytes() + mass.tobytes()
        block(f, header + bytes(256 - len(header)))
        block(f, pos.tobytes())                                   # POS
        block(f, np.zeros((n, 3), np.float32).tobytes())          # VEL
        block(f, np.arange(n, dtype=np.int32).tobytes())          # ID
        if not masses_in_header:
            block(f, np.ones(n, np.float32).tobytes())            # MASS (only if header mass == 0)
        block(f, np.zeros(n, np.float32).tobytes())               # U
        block(f, np.ones(n, np.float32).tobytes())                # RHO
        block(f, hsml.tobytes())                                  # HSML


def read_gadget(fname):
    """Returns spheres [N_gas, 4] float32 = (x, y, z, hsml) -- read_gadget.cuh:69-159."""
    with open(fname, "rb") as f:
        raw = np.frombuffer(f.read(), np.uint8)
    pos = 0

    def take_block():
        nonlocal pos
        nbytes = int(raw[pos:pos + 4].view(np.int32)[0])
        data = raw[pos + 4: pos + 4 + nbytes]
        pos += 8 + nbytes
        return data

    header = take_block()
    npart = header[:24].view(np.int32)
    mass = header[24:72].view(np.float64)
    n_gas = int(npart[0])
    if n_gas == 0:
        raise RuntimeError("Gadget file %s has no gas particles!" % fname)
    n_withmass = int(sum(int(npart[i]) for i in range(6) if mass[i] == 0))
    p = take_block().view(np.float32).reshape(-1, 3)[:n_gas]
    take_block()                 # VEL
    take_block()                 # ID
    if n_withmass > 0:
        take_block()             # MASS
    take_block()                 # U
    take_block()                 # RHO
    h = take_block().view(np.float32)[:n_gas]
    out = np.empty((n_gas, 4), np.float32)
    out[:, :3] = p
    out[:, 3] = h
    return out
