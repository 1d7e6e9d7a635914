"""temporary: cycle counters per section of spdh_rowwave's step (apply, build, run, then restore the two files)"""
p='/root/repo/spaln_amd/csrc/spdp_h_rowwave.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b,1)
rep("            for (int S = S_begin; S <= S_end; ++S) {\n                int v = S - m;","            long long tacc[8] = {0,0,0,0,0,0,0,0}; long long tlast = __builtin_readcyclecounter();\n#define TICK(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }\n            for (int S = S_begin; S <= S_end; ++S) {\n                int v = S - m;\n                TICK(0)")
rep("                const int n = RV(v);\n                const bool on = any && act && v >= v0 && v <= v9;\n                if (__ballot(on) == 0) continue;","                TICK(1)\n                const int n = RV(v);\n                const bool on = any && act && v >= v0 && v <= v9;\n                if (__ballot(on) == 0) continue;")
rep("                // ---- acceptor: the candidates of this column's phase(s) may raise the state they left from\n","                TICK(2)\n                // ---- acceptor: the candidates of this column's phase(s) may raise the state they left from\n")
rep("                // ---- the cell takes the best state\n","                TICK(3)\n                // ---- the cell takes the best state\n")
rep("                // ---- donor: the states of this cell enter the candidate list(s)","                TICK(4)\n                // ---- donor: the states of this cell enter the candidate list(s)")
rep("                if (on) {\n                    lds_put(e, 0, h);\n                    lds_put(e, 1, f);","                TICK(5)\n                if (on) {\n                    lds_put(e, 0, h);\n                    lds_put(e, 1, f);")
rep("            WAVE_SYNC();\n            for (int e = res_lo + lane; e < res_hi; e += 64) {\n                const int q = e & (RING - 1);\n#pragma unroll\n                for (int a = 0; a < 2 * NF; ++a) gst<PIPE>(G(a) + e, L[a][q]);\n            }\n        }\n        if (UDH && PIPE) {","            TICK(6)\n            if (PIPE && lane == 0) for (int i = 0; i < 7; ++i) atomicAdd((unsigned long long*) (A.pipe + A.pipe_ticket + 2) + i, (unsigned long long) tacc[i]);\n            WAVE_SYNC();\n            for (int e = res_lo + lane; e < res_hi; e += 64) {\n                const int q = e & (RING - 1);\n#pragma unroll\n                for (int a = 0; a < 2 * NF; ++a) gst<PIPE>(G(a) + e, L[a][q]);\n            }\n        }\n        if (UDH && PIPE) {")
open(p,'w').write(s)
p='/root/repo/spaln_amd/csrc/spdp_h_api.cpp'
s=open(p).read()
rep("    pp.words = (probs.size() * (size_t) pp.stride + 2 + 1) & ~(size_t) 1;\n    pp.d = (int*) pool.get(slot, sizeof(int) * (pp.words + pp.items.size()));\n    if (!pp.d) { ctx->err = \"device allocation failed (tile pipeline of the scalar aa x genome engines)\"; return -1; }","    pp.words = (probs.size() * (size_t) pp.stride + 2 + 32 + 1) & ~(size_t) 1;\n    pp.d = (int*) pool.get(slot, sizeof(int) * (pp.words + pp.items.size()));\n    if (!pp.d) { ctx->err = \"device allocation failed (tile pipeline of the scalar aa x genome engines)\"; return -1; }")
rep("    int mark[2] = {0, 0};\n    HIPCHK(spdp_copy_sync(mark, pp.d + (size_t) n_probs * pp.stride, sizeof mark, hipMemcpyDeviceToHost, ctx->stream));","    int mark[2] = {0, 0};\n    HIPCHK(spdp_copy_sync(mark, pp.d + (size_t) n_probs * pp.stride, sizeof mark, hipMemcpyDeviceToHost, ctx->stream));\n    { unsigned long long t[7]; HIPCHK(spdp_copy_sync(t, pp.d + (size_t) n_probs * pp.stride + 2, sizeof t, hipMemcpyDeviceToHost, ctx->stream)); fprintf(stderr, \"[timing] loop-top %llu refill %llu loads+recurrence %llu acceptor %llu best-state %llu donor %llu put %llu (M cycles)\\n\", t[0]>>20, t[1]>>20, t[2]>>20, t[3]>>20, t[4]>>20, t[5]>>20, t[6]>>20); }")
open(p,'w').write(s)
