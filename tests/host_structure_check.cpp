// Sanitizer harness of the host half of the structure build (mc_slam_amd/csrc/vba_host_structure.h) and of the problem file
// reader (vba_problem_io.h): plain C++, built by tests/test_host_structure.py with g++ -fsanitize=address,undefined.
//   host_structure_check <file.vbap>...   -> one line per file: "ok <summary>" or "error <message>"
#include "../mc_slam_amd/csrc/vba_host_structure.h"
#include "../mc_slam_amd/csrc/vba_problem_io.h"

#include <cstdint>

static uint64_t fnv(const std::vector<int>& v, uint64_t h = 1469598103934665603ull) {
    for (int x : v) { h ^= (uint32_t)x; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    for (int a = 1; a < argc; a++) {
        vba_problem* P = nullptr;
        const int rc = vba_problem_load(argv[a], &P);
        if (rc) { printf("error load %d\n", rc); continue; }
        // twice: with the orders 0 / 1 only, then with the two-sided order as a candidate (what the library asks for); the second is printed
        bool ok_all = true, failed = false;
        vba_host::Structure st;
        std::string err;
        for (int two = 0; two < 2 && !failed; two++) {
        st = vba_host::Structure();
        if (vba_host::build_structure(P, st, err, two != 0)) { printf("error %s\n", err.c_str()); failed = true; break; }
        long long mask_bits = 0;
        for (unsigned long long m : st.lmask) mask_bits += __builtin_popcountll(m);
        const int nf = P->n_kf_free, npairs = nf * (nf + 1) / 2;
        // internal consistency the device side relies on
        bool ok = (int)st.pair_a.size() == npairs && (int)st.pair_mask.size() == npairs && (int)st.pimu_begin.size() == npairs + 1 &&
                  st.lmask.size() == (size_t)P->n_pt * st.mwords && st.kl_begin.size() == st.pan.size() + st.step_begin.size();
        for (size_t i = 0; i + 1 < st.pimu_begin.size(); i++) ok = ok && st.pimu_begin[i] <= st.pimu_begin[i + 1];
        ok = ok && (size_t)st.pimu_begin.back() * 2 == st.pimu.size();
        if (P->variant == VBA_VARIANT_PRV_IDP) {   // the k_lin2 runs cover every landmark and edge exactly once, within the limits
            int p = 0;
            for (size_t i = 0; i + 3 < st.linblk.size() + 1 && i < st.linblk.size(); i += 4) {
                ok = ok && st.linblk[i] == p && st.linblk[i + 1] > p && st.linblk[i + 1] - p <= 64 && st.linblk[i + 3] - st.linblk[i + 2] <= 256 &&
                     st.linblk[i + 2] == P->pt_obs_begin[p] && st.linblk[i + 3] == P->pt_obs_begin[st.linblk[i + 1]];
                p = st.linblk[i + 1];
            }
            ok = ok && p == P->n_pt;
            // the run records of the reference-keyframe terms: per workgroup the maximal runs of consecutive landmarks with one reference
            // keyframe, ids in workgroup order; per keyframe the ids of its runs, ascending, every run listed exactly once
            const int nblk = (int)(st.linblk.size() / 4);
            ok = ok && (int)st.prun0.size() == nblk + 1 && (int)st.pref_begin.size() == P->n_kf + 1;
            int id = 0;
            std::vector<int> ref_of;
            for (int lb = 0; lb < nblk && ok; lb++) {
                ok = ok && st.prun0[lb] == id;
                for (int q = st.linblk[4 * lb]; q < st.linblk[4 * lb + 1]; q++)
                    if (q == st.linblk[4 * lb] || P->pt_ref_kf[q] != P->pt_ref_kf[q - 1]) { ref_of.push_back(P->pt_ref_kf[q]); id++; }
            }
            ok = ok && st.prun0.back() == id && (int)st.pref_list.size() == id && id <= P->n_pt && st.pref_begin.back() == id;
            std::vector<int> seen(id, 0);
            for (int k = 0; k < P->n_kf && ok; k++)
                for (int m = st.pref_begin[k]; m < st.pref_begin[k + 1]; m++) {
                    const int r = st.pref_list[m];
                    ok = ok && r >= 0 && r < id && ref_of[r] == k && !seen[r] && (m == st.pref_begin[k] || st.pref_list[m - 1] < r);
                    if (ok) seen[r] = 1;
                }
        }
        {   // chain columns: row J < nc of L has no tile left of (J, J-1); the tables of the chain kernels agree with the tile lists
            const int nb = (int)st.step_begin.size() - 1;
            ok = ok && st.nc >= 0 && st.nc < std::max(nb, 1) && (int)st.chain_tab.size() == 4 * st.nc && st.cu.size() % 4 == 0;
            for (int J = 0; J < st.nc && ok; J++) {
                unsigned long long mask = 0;
                int ride = 0;
                for (int i = st.pan_begin[J]; i < st.pan_begin[J + 1]; i++) {
                    const int I = st.pan[i];
                    if (I == J + 1 && J + 1 < st.nc) ride = 1;
                    else { ok = ok && I >= st.nc && I - st.nc < 64; mask |= 1ull << ((I - st.nc) & 63); }
                }
                ok = ok && (unsigned)st.chain_tab[4 * J] == (unsigned)(mask & 0xffffffffull) && (unsigned)st.chain_tab[4 * J + 1] == (unsigned)(mask >> 32) && st.chain_tab[4 * J + 2] == ride;
                const int ent = st.pan_begin[J] + J;   // the k list of the diagonal entry: empty or {J - 1}
                const int nk = st.kl_begin[ent + 1] - st.kl_begin[ent];
                ok = ok && (nk == 0 || (nk == 1 && st.klist[st.kl_begin[ent]] == J - 1));
            }
            for (size_t q = 0; q + 3 < st.cu.size() && ok; q += 4) {
                const int I = st.cu[q] >> 16, J = st.cu[q] & 0xffff;
                ok = ok && J >= st.nc && I >= J && st.cu[q + 1] < st.cu[q + 2];
                for (int e = st.cu[q + 1]; e < st.cu[q + 2]; e++) ok = ok && st.klist[e] < st.nc;
            }
        }
        {   // the positions of the variables under the chosen order: distinct, inside the rows the structure reports
            const int pdim = (P->variant == VBA_VARIANT_SE3_XYZ) ? 6 : 15, nb = (int)st.step_begin.size() - 1;
            ok = ok && st.nS == nb * VBA_NB && (two || st.order != 2) && vba_host::order_rows(st.order, pdim, nf) <= st.nS;
            std::vector<char> seen((size_t)std::max(st.nS, 1), 0);
            for (int k = 0; k < nf && ok; k++)
                for (int r = 0; r < pdim && ok; r++) {
                    const int v = vba_host::vpos_host(st.order, pdim, nf, k, r);
                    ok = ok && v >= 0 && v < st.nS && !seen[v];
                    if (ok) seen[v] = 1;
                }
            // the split of the chain: no tile couples column nc_split - 1 to column nc_split
            ok = ok && st.nc_split >= 0 && (st.nc_split == 0 || (st.nc_split < st.nc && st.chain_tab[4 * (st.nc_split - 1) + 2] == 0));
            if (st.order == 2) ok = ok && (st.nc == 0 || st.nc_split > 0);
        }
        ok_all = ok_all && ok;
        if (!two) continue;
        ok = ok_all;
        printf("%s order %d mwords %d item_cap %lld mask_bits %lld tiles %zu klist %zu pimu %zu pan %zu nc %d split %d h_tiles %llx h_mask %llx\n", ok ? "ok" : "error inconsistent",
               st.order, st.mwords, st.item_cap, mask_bits, st.tpairs.size(), st.klist.size(), st.pimu.size() / 2, st.pan.size(), st.nc, st.nc_split,
               (unsigned long long)fnv(st.tpairs), (unsigned long long)fnv(st.pair_mask));
        }
        vba_problem_free(P);
    }
    return 0;
}
