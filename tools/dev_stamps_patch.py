"""Development aid: patch in-kernel cycle stamps (s_memtime at phase boundaries of k_mfma, written to the idle
128-float slot at the end of each wave's stash) into COPIES of the kernel sources, for a tools/dev_build.sh build:
    python tools/dev_stamps_patch.py apply    # patches pde_amd/csrc/{kernel_mfma.hpp,pinnjet.hip} in place (backups in /tmp)
    tools/dev_build.sh stamp && (GPU) python tools/dev_stamps_read.py
    python tools/dev_stamps_patch.py restore
Never commit the patched sources."""
import shutil
import sys

K = "pde_amd/csrc/kernel_mfma.hpp"
H = "pde_amd/csrc/pinnjet.hip"


def apply():
    shutil.copy(K, "/tmp/kernel_mfma.clean.hpp")
    shutil.copy(H, "/tmp/pinnjet.clean.hip")
    s = open(K).read()
    def rep(old, new):
        nonlocal s
        assert old in s, old
        s = s.replace(old, new, 1)
    rep('''    float *const stage = lds + L.stage + wv * STG * STAGE_TILE;''', '''    int stamp_idx = 0;
#define STAMP(id) do { if (tile >= nwaves && tile < 2 * nwaves) { unsigned long long tm = __builtin_amdgcn_s_memtime(); if (lane0 == 0) { st[XT_OFF + 2 * stamp_idx] = __int_as_float((int)(tm & 0xffffffffu)); st[XT_OFF + 2 * stamp_idx + 1] = __int_as_float(id); } ++stamp_idx; } } while (0)
    float *const stage = lds + L.stage + wv * STG * STAGE_TILE;''')
    rep('''        // ---- first layer (2 -> W), VALU ----''', '''        stamp_idx = 0; STAMP(0);
        // ---- first layer (2 -> W), VALU ----''')
    rep('''        // ---- hidden -> hidden layers on the matrix cores, one jet component at a time, in place ----''', '''        STAMP(1);
        // ---- hidden -> hidden layers on the matrix cores, one jet component at a time, in place ----''')
    rep('''            // A = z of this layer.  Last tanh layer: z stays, y feeds the output sums.  Otherwise: stash z''', '''            STAMP(10 + l);
            // A = z of this layer.  Last tanh layer: z stays, y feeds the output sums.  Otherwise: stash z''')
    rep('''        // ---- output layer (W -> nout), VALU + cross-quad reduction ----''', '''        STAMP(2);
        // ---- output layer (W -> nout), VALU + cross-quad reduction ----''')
    rep('''        // ---- output layer: bias gradient ----''', '''        STAMP(3);
        // ---- output layer: bias gradient ----''')
    rep('''        // ---- Linear layers l = NH-1 .. 1 (hidden -> hidden), each followed by the tanh layer below it ----''', '''        STAMP(4);
        // ---- Linear layers l = NH-1 .. 1 (hidden -> hidden), each followed by the tanh layer below it ----''')
    rep('''            // Phase B: Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p], one jet component at a time, in place.''', '''            STAMP(20 + l);
            // Phase B: Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p], one jet component at a time, in place.''')
    rep('''            // tanh layer hl = l-1: A/At = adjoint of y_hl -> adjoint of z_hl, in place''', '''            STAMP(30 + l);
            // tanh layer hl = l-1: A/At = adjoint of y_hl -> adjoint of z_hl, in place''')
    rep('''        // ---- first layer gradients from A/At = adjoint of z_0 ----''', '''        STAMP(5);
        // ---- first layer gradients from A/At = adjoint of z_0 ----''')
    rep('''    }

    PINN_REFRESH();
    if (MODE != MODE_FWD) {
        if (MODE == MODE_LOSS) {''', '''        STAMP(6);
    }

    PINN_REFRESH();
    if (MODE != MODE_FWD) {
        if (MODE == MODE_LOSS) {''')
    rep('''        for (int l = NH - 1; l >= 1; --l) {
            PINN_REFRESH();''', '''        for (int l = NH - 1; l >= 1; --l) {
            STAMP(40 + l);
            PINN_REFRESH();''')
    open(K, "w").write(s)
    h = open(H).read()
    old = "int pinn_reserve(pinn_handle *h, int64_t n_max)"
    assert old in h
    h = h.replace(old, '''int pinn_debug_read_jets(pinn_handle *h, void *dst, int64_t offset_bytes, int64_t bytes)
{
    return hipMemcpy(dst, (char *)h->jets + offset_bytes, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

''' + old, 1)
    open(H, "w").write(h)


def restore():
    shutil.copy("/tmp/kernel_mfma.clean.hpp", K)
    shutil.copy("/tmp/pinnjet.clean.hip", H)


if __name__ == "__main__":
    {"apply": apply, "restore": restore}[sys.argv[1]]()
