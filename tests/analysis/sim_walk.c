/* Analysis only (not product, not a test): simulates the packet walk of csrc/bvh.hip's phase 2 on a tree built by
 * the oracle, with and without "leaf blocks" (a hit node covering <= B leaves is tested leaf by leaf instead of being
 * descended into).  Counts steps per packet.  Built and driven by tests/analysis/sim_packet_walk.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t parent, right_edge, data[2]; } node_t;

/* bounds: (2n-1) x 2 x 4 floats.  out[0..5] = steps, descents, leaf steps, leaf hits(lane-pairs), blocks, block leaves */
void sim_walk(const node_t *nodes, const float *bounds, uint32_t n, int B, uint64_t *out) {
    const uint32_t nn = 2 * n - 1, leaf0 = n - 1, END = 0xFFFFFFFFu;
    uint32_t *skip = malloc(4ull * nn), *lo = malloc(4ull * nn), *cnt = malloc(4ull * nn), *stack = malloc(4ull * 128);
    /* preorder from the root: skip links; post-order for (lo, cnt) via the Karras layout: node range = [min(i, other), right_edge] */
    skip[0] = END;
    int sp = 0; stack[sp++] = 0;
    /* iterative DFS; depth <= 64 + ... use heap stack sized nn to be safe */
    free(stack); stack = malloc(4ull * nn); sp = 0; stack[sp++] = 0;
    while (sp) {
        uint32_t x = stack[--sp];
        if (x >= leaf0) continue;
        uint32_t a = nodes[x].data[0], b = nodes[x].data[1];
        skip[a] = b; skip[b] = skip[x];
        stack[sp++] = a; stack[sp++] = b;
    }
    /* ranges: leaves first, then internal nodes bottom-up by repeated passes is O(depth*n); instead compute via right_edge and
     * leftmost leaf by walking down-left (cheap enough: average depth ~ log n) */
    for (uint32_t x = 0; x < nn; x++) {
        uint32_t y = x;
        while (y < leaf0) y = nodes[y].data[0];
        lo[x] = y - leaf0;
        cnt[x] = (x >= leaf0 ? (x - leaf0) : nodes[x].right_edge) - lo[x] + 1;
    }
    memset(out, 0, 8 * 8);
    const uint32_t npackets = (n + 63) / 64;
    for (uint32_t p = 0; p < npackets; p++) {
        const uint32_t q0 = p * 64, q1 = (q0 + 64 < n ? q0 + 64 : n);
        uint32_t idx = skip[leaf0 + q1 - 1];
        while (idx != END) {
            const float *r = bounds + 8ull * idx;
            int any = 0; uint64_t hits = 0;
            for (uint32_t q = q0; q < q1; q++) {
                const float *b = bounds + 8ull * (leaf0 + q);
                int h = b[4] > r[0] && b[0] < r[4] && b[5] > r[1] && b[1] < r[5] && b[6] > r[2] && b[2] < r[6];
                any |= h; hits += h;
            }
            out[0]++;
            uint32_t next = skip[idx];
            if (any) {
                if (idx >= leaf0) { out[2]++; out[3] += hits; }
                else if (B > 0 && cnt[idx] <= (uint32_t)B) { out[4]++; out[5] += cnt[idx]; }
                else { out[1]++; next = nodes[idx].data[0]; }
            }
            idx = next;
        }
    }
    free(skip); free(lo); free(cnt); free(stack);
}

/* K interleaved walk contexts per packet, in lockstep rounds (every active context fetches one record per round, the
 * fetches of a round overlap).  A context walks a range [cur, term) of the skip chain; when it fetches a node whose
 * skip link is not its terminator and another context is idle, it keeps the node's subtree and hands [skip, term) over.
 * out[0] = rounds, out[1] = fetches, out[2] = rounds in which only one context fetched.  Leaf blocks (B) as in sim_walk:
 * a block is tested inside the round of the context that met it. */
void sim_walk_ctx(const node_t *nodes, const float *bounds, uint32_t n, int B, int K, uint64_t *out) {
    const uint32_t nn = 2 * n - 1, leaf0 = n - 1, END = 0xFFFFFFFFu;
    uint32_t *skip = malloc(4ull * nn), *lo = malloc(4ull * nn), *cnt = malloc(4ull * nn), *stack = malloc(4ull * nn);
    int sp = 0;
    skip[0] = END; stack[sp++] = 0;
    while (sp) {
        uint32_t x = stack[--sp];
        if (x >= leaf0) continue;
        uint32_t a = nodes[x].data[0], b = nodes[x].data[1];
        skip[a] = b; skip[b] = skip[x];
        stack[sp++] = a; stack[sp++] = b;
    }
    for (uint32_t x = 0; x < nn; x++) {
        uint32_t y = x;
        while (y < leaf0) y = nodes[y].data[0];
        lo[x] = y - leaf0;
        cnt[x] = (x >= leaf0 ? (x - leaf0) : nodes[x].right_edge) - lo[x] + 1;
    }
    memset(out, 0, 8 * 8);
    const uint32_t npackets = (n + 63) / 64;
    for (uint32_t p = 0; p < npackets; p++) {
        const uint32_t q0 = p * 64, q1 = (q0 + 64 < n ? q0 + 64 : n);
        uint32_t cur[8], term[8]; int act[8];
        for (int k = 0; k < K; k++) act[k] = 0;
        cur[0] = skip[leaf0 + q1 - 1]; term[0] = END; act[0] = cur[0] != END;
        for (;;) {
            int nact = 0;
            for (int k = 0; k < K; k++) nact += act[k];
            if (!nact) break;
            out[0]++; out[1] += nact; if (nact == 1) out[2]++;
            int was[8];
            for (int k = 0; k < K; k++) was[k] = act[k];
            for (int k = 0; k < K; k++) {
                if (!was[k]) continue;
                const uint32_t idx = cur[k];
                const float *r = bounds + 8ull * idx;
                int any = 0;
                for (uint32_t q = q0; q < q1 && !any; q++) {
                    const float *b = bounds + 8ull * (leaf0 + q);
                    any = b[4] > r[0] && b[0] < r[4] && b[5] > r[1] && b[1] < r[5] && b[6] > r[2] && b[2] < r[6];
                }
                uint32_t next = skip[idx];
                const int descend = any && idx < leaf0 && !(B > 0 && cnt[idx] <= (uint32_t)B);
                if (descend && next != term[k]) {            /* hand the continuation over to an idle context */
                    for (int j = 0; j < K; j++)
                        if (!act[j]) { act[j] = 1; cur[j] = next; term[j] = term[k]; term[k] = next; break; }
                }
                if (descend) next = nodes[idx].data[0];
                cur[k] = next;
                if (next == term[k]) act[k] = 0;
            }
        }
    }
    free(skip); free(lo); free(cnt); free(stack);
}

/* How would a packet's walk split in two at the j-th node of its top-level chain (the chain skip(last leaf), skip(that), ...)?
 * out[j][0] = steps spent in the first j top-level subtrees, out[j][1] = steps in the rest, summed over the packets;
 * out[j][2] / out[j][3] = the maximum of either part over the packets.  j = 1 .. 15. */
void sim_walk_split(const node_t *nodes, const float *bounds, uint32_t n, uint64_t (*out)[4], uint32_t *per_packet /* [npackets][16]: steps by top-level subtree */) {
    const uint32_t nn = 2 * n - 1, leaf0 = n - 1, END = 0xFFFFFFFFu;
    uint32_t *skip = malloc(4ull * nn), *stack = malloc(4ull * nn);
    int sp = 0;
    skip[0] = END; stack[sp++] = 0;
    while (sp) {
        uint32_t x = stack[--sp];
        if (x >= leaf0) continue;
        uint32_t a = nodes[x].data[0], b = nodes[x].data[1];
        skip[a] = b; skip[b] = skip[x];
        stack[sp++] = a; stack[sp++] = b;
    }
    const uint32_t npackets = (n + 63) / 64;
    memset(out, 0, 16 * 4 * 8);
    for (uint32_t p = 0; p < npackets; p++) {
        const uint32_t q0 = p * 64, q1 = (q0 + 64 < n ? q0 + 64 : n);
        uint32_t top = skip[leaf0 + q1 - 1];
        uint32_t steps[64]; int k = 0;
        memset(steps, 0, sizeof(steps));
        while (top != END) {
            const uint32_t term = skip[top];
            uint32_t idx = top, cnt = 0;
            while (idx != term) {
                const float *r = bounds + 8ull * idx;
                int any = 0;
                for (uint32_t q = q0; q < q1 && !any; q++) {
                    const float *b = bounds + 8ull * (leaf0 + q);
                    any = b[4] > r[0] && b[0] < r[4] && b[5] > r[1] && b[1] < r[5] && b[6] > r[2] && b[2] < r[6];
                }
                cnt++;
                idx = (any && idx < leaf0) ? nodes[idx].data[0] : skip[idx];
            }
            if (k < 64) steps[k] = cnt;
            k++;
            top = term;
        }
        uint32_t total = 0;
        for (int i = 0; i < 64; i++) total += steps[i];
        uint32_t pre = 0;
        for (int j = 1; j < 16; j++) {
            pre += steps[j - 1];
            out[j][0] += pre; out[j][1] += total - pre;
            if (pre > out[j][2]) out[j][2] = pre;
            if (total - pre > out[j][3]) out[j][3] = total - pre;
        }
        if (per_packet) for (int i = 0; i < 16; i++) per_packet[16ull * p + i] = i < 15 ? steps[i] : total;
    }
    free(skip); free(stack);
}
