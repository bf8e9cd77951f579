// cx_tiling.h — host-side graph clustering for the two-sweep kernel (cx_tiles.hip): pure C++, no HIP, so that it can be
// exercised on a CPU (tools/tile_lab.cpp, tests/test_tiling_cpu.py).
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace cx {

// recursive bisection along breadth-first levels: pieces of at most `cap` variables, each a contiguous run of `order`
inline void bisect(const std::vector<int32_t> &adj_off, const std::vector<int32_t> &adj, int32_t nv, int cap, std::vector<int32_t> &order,
            std::vector<int32_t> &piece_end) {
    order.resize(nv);
    std::iota(order.begin(), order.end(), 0);
    std::vector<int32_t> piece(nv, 0), stamp(nv, -1), queue(nv), tmp;
    struct Range { int32_t lo, hi; };
    std::vector<Range> todo{{0, nv}};
    int32_t next_piece = 1, bfs_id = 0;
    std::vector<Range> done;
    auto bfs = [&](int32_t lo, int32_t hi, int32_t start_pos, int32_t pid) {
        // breadth-first order of the piece order[lo..hi) from order[start_pos] (restarting in unvisited parts); result in queue[0..n)
        const int32_t id = bfs_id++;
        int32_t head = 0, tail = 0, scan = lo;
        auto push = [&](int32_t v) { stamp[v] = id; queue[tail++] = v; };
        push(order[start_pos]);
        const int32_t n = hi - lo;
        while (tail < n || head < tail) {
            if (head == tail) {   // disconnected remainder
                while (stamp[order[scan]] == id) scan++;
                push(order[scan]);
            }
            const int32_t v = queue[head++];
            for (int32_t e = adj_off[v]; e < adj_off[v + 1]; e++) {
                const int32_t w = adj[e];
                if (piece[w] == pid && stamp[w] != id) push(w);
            }
        }
        return n;
    };
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        const int32_t n = r.hi - r.lo;
        if (n <= cap) { done.push_back(r); continue; }
        const int32_t pid = piece[order[r.lo]];
        bfs(r.lo, r.hi, r.lo, pid);
        const int32_t far = queue[n - 1];               // pseudo-peripheral vertex: the last one reached
        // second pass from the far end: its level order is what the piece is cut along
        int32_t far_pos = r.lo;
        for (int32_t i = r.lo; i < r.hi; i++) if (order[i] == far) { far_pos = i; break; }
        bfs(r.lo, r.hi, far_pos, pid);
        // the piece will end as m = ceil(n / cap) tiles: the left part takes floor(m / 2) of them, so final tiles stay close to cap
        const int64_t m = (n + cap - 1) / cap;
        const int32_t nleft = (int32_t)(((int64_t)n * (m / 2)) / m);
        for (int32_t i = 0; i < n; i++) order[r.lo + i] = queue[i];
        const int32_t pl = next_piece++, pr = next_piece++;
        for (int32_t i = 0; i < nleft; i++) piece[order[r.lo + i]] = pl;
        for (int32_t i = nleft; i < n; i++) piece[order[r.lo + i]] = pr;
        todo.push_back({r.lo + nleft, r.hi});
        todo.push_back({r.lo, r.lo + nleft});
    }
    std::sort(done.begin(), done.end(), [](const Range &a, const Range &b) { return a.lo < b.lo; });
    piece_end.clear();
    for (const Range &r : done) piece_end.push_back(r.hi);
}



// one more cut of a single piece (a tile whose two-hop ring came out too large: a thin or ragged cluster): breadth-first order
// of `members` inside the piece from a pseudo-peripheral vertex, halves returned in place (first `return value` entries = left)
inline int32_t split_piece(const std::vector<int32_t> &adj_off, const std::vector<int32_t> &adj, std::vector<int32_t> &members,
                           std::vector<int32_t> &in_piece /* scratch, all -1 on entry and on exit */) {
    const int32_t n = (int32_t)members.size();
    for (int32_t v : members) in_piece[v] = 0;
    std::vector<int32_t> queue;
    queue.reserve(n);
    auto bfs = [&](int32_t start, int32_t tag) {
        queue.clear();
        size_t head = 0, scan = 0;
        in_piece[start] = tag; queue.push_back(start);
        while ((int32_t)queue.size() < n || head < queue.size()) {
            if (head == queue.size()) {
                while (in_piece[members[scan]] == tag) scan++;
                in_piece[members[scan]] = tag; queue.push_back(members[scan]);
            }
            const int32_t v = queue[head++];
            for (int32_t e = adj_off[v]; e < adj_off[v + 1]; e++) {
                const int32_t w = adj[e];
                if (in_piece[w] >= 0 && in_piece[w] != tag) { in_piece[w] = tag; queue.push_back(w); }
            }
        }
    };
    bfs(members[0], 1);
    const int32_t far = queue.back();
    bfs(far, 2);
    for (int32_t v : members) in_piece[v] = -1;
    members = queue;
    return n / 2;
}

}  // namespace cx
