// oracle/amt.cpp — TEST INFRASTRUCTURE (see amt.hpp for provenance).
#include "amt.hpp"

#include <malloc.h>
#include <omp.h>

#include <cstdlib>
#include <memory>

namespace orc {

namespace {

struct Node {
    uint32_t width = 0;
    std::vector<uint8_t> bmap;
    bool is_link = false;            // links non-empty
    std::vector<Cid> links;          // compacted
    std::vector<ValueLoc> values;    // compacted
    bool bit(uint32_t i) const { return bmap[i / 8] & (1u << (i % 8)); }
    uint32_t rank(uint32_t i) const {  // set bits below i
        uint32_t r = 0;
        for (uint32_t k = 0; k < i; ++k) r += bit(k) ? 1 : 0;
        return r;
    }
};

// Decode `[bmap, links, values]` at r (CollapsedNode) and expand-check it.
Node read_node(Reader& r, const Bytes* block, uint32_t bw, const ValueChecker& check) {
    Node nd;
    nd.width = 1u << bw;
    r.expect_array(3);
    nd.bmap = r.read_bytes_vec();
    const uint64_t nlinks = r.read_array();
    for (uint64_t i = 0; i < nlinks; ++i) nd.links.push_back(read_cid(r));
    const uint64_t nvals = r.read_array();
    for (uint64_t i = 0; i < nvals; ++i) {
        ValueLoc v;
        v.block = block;
        v.off = r.pos;
        check(r);
        v.len = r.pos - v.off;
        nd.values.push_back(v);
    }
    if (!nd.links.empty() && !nd.values.empty()) decode_err("AMT node has both links and values");
    if (nd.bmap.size() != (nd.width + 7) / 8) decode_err("AMT bitmap has the wrong length");
    uint32_t pop = 0;
    for (uint32_t i = 0; i < nd.width; ++i) pop += nd.bit(i) ? 1 : 0;
    // bits of the last byte beyond `width` (bit_width < 3) are ignored, as the crate's loop does
    nd.is_link = !nd.links.empty();
    const size_t have = nd.is_link ? nd.links.size() : nd.values.size();
    if (have != pop) decode_err("AMT node entry count does not match its bitmap");
    return nd;
}

uint64_t nodes_for_height(uint32_t bw, uint64_t height) {
    const uint64_t shift = uint64_t(bw) * height;
    return shift >= 64 ? UINT64_MAX : (1ull << shift);
}

Node load_child(const Blockstore& bs, const Cid& c, uint32_t bw, const ValueChecker& check) {
    const Bytes& raw = must_get(bs, c, "AMT node");
    Reader r(raw);
    Node nd = read_node(r, &raw, bw, check);
    r.finish();
    return nd;
}

bool node_get(const Blockstore& bs, const Node& nd, uint64_t height, uint32_t bw, uint64_t i,
              const ValueChecker& check, ValueLoc& loc) {
    if (!nd.is_link) {
        // Node::Leaf: `vals.get(i)`
        if (i >= nd.width || !nd.bit(uint32_t(i))) return false;
        loc = nd.values[nd.rank(uint32_t(i))];
        return true;
    }
    if (height == 0) decode_err("AMT link node at height 0");
    const uint64_t span = nodes_for_height(bw, height);
    const uint64_t sub = i / span;
    if (sub >= nd.width || !nd.bit(uint32_t(sub))) return false;
    Node child = load_child(bs, nd.links[nd.rank(uint32_t(sub))], bw, check);
    return node_get(bs, child, height - 1, bw, i % span, check, loc);
}

void node_for_each(const Blockstore& bs, const Node& nd, uint64_t height, uint32_t bw, uint64_t base,
                   const ValueChecker& check, const std::function<void(uint64_t, const ValueLoc&)>& f) {
    if (!nd.is_link) {
        uint32_t k = 0;
        for (uint32_t i = 0; i < nd.width; ++i)
            if (nd.bit(i)) f(base + i, nd.values[k++]);
        return;
    }
    if (height == 0) decode_err("AMT link node at height 0");
    const uint64_t span = nodes_for_height(bw, height);
    uint32_t k = 0;
    for (uint32_t i = 0; i < nd.width; ++i) {
        if (!nd.bit(i)) continue;
        Node child = load_child(bs, nd.links[k++], bw, check);
        node_for_each(bs, child, height - 1, bw, base + uint64_t(i) * span, check, f);
    }
}

Node root_node(const AmtRoot& root, const ValueChecker& check) {
    Reader r(*root.block);
    r.pos = root.node_off;
    return read_node(r, root.block, root.bit_width, check);
}

}  // namespace

AmtRoot amt_load(const Blockstore& bs, const Cid& c, int version, const ValueChecker& check) {
    const Bytes& raw = must_get(bs, c, "AMT root");
    Reader r(raw);
    AmtRoot root;
    root.block = &raw;
    if (version == 0) {
        r.expect_array(3);
        root.bit_width = 3;
    } else {
        r.expect_array(4);
        const uint64_t bw = r.read_uint();
        if (bw < 1 || bw > kAmtMaxBitWidth) decode_err("AMT bit width out of the supported range");
        root.bit_width = uint32_t(bw);
    }
    root.height = r.read_uint();
    root.count = r.read_uint();
    root.node_off = r.pos;
    (void)read_node(r, &raw, root.bit_width, check);
    r.finish();
    // `if root.height > MAX_HEIGHT { return Err(MaxHeight) }`, MAX_HEIGHT = 64 / bit_width
    if (root.height > 64 / root.bit_width) decode_err("AMT height above the maximum");
    return root;
}

bool amt_get(const Blockstore& bs, const AmtRoot& root, uint64_t index, const ValueChecker& check, ValueLoc& loc) {
    // `if i > MAX_INDEX` (u64::MAX - 1) → Err(OutOfRange)
    if (index == UINT64_MAX) throw Err(IPCFP_ST_ERR, "AMT index out of range");
    if (index >= nodes_for_height(root.bit_width, root.height + 1)) return false;
    Node nd = root_node(root, check);
    return node_get(bs, nd, root.height, root.bit_width, index, check, loc);
}

void amt_for_each(const Blockstore& bs, const AmtRoot& root, const ValueChecker& check,
                  const std::function<void(uint64_t, const ValueLoc&)>& f) {
    Node nd = root_node(root, check);
    node_for_each(bs, nd, root.height, root.bit_width, 0, check, f);
}


// glibc grows a thread's malloc arena one page per mprotect() call, and every mprotect takes the process-wide
// mmap lock exclusively: a few hundred threads that all allocate results (store blocks, hash-map nodes, item
// lists) serialise in the kernel and the "all cores" baseline runs SLOWER than one thread (measured on the
// 256-processor GPU box: store build 1.0 s vs 0.45 s, execution order 3.5 s vs 1.1 s).  Two remedies, both
// applied before a parallel phase: the arenas never give memory back (no trim) and big requests stay inside
// them, and every thread grows its arena once, by one large step, instead of ten thousand small ones.
static void pregrow_arenas() {
    static bool tuned = false;
    if (!tuned) {
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
        tuned = true;
    }
#pragma omp parallel
    {
        void* p = std::malloc(size_t(24) << 20);  // one mprotect of 24 MB (pages are still touched lazily)
        if (p) *static_cast<volatile char*>(p) = 0;
        std::free(p);
    }
}

int use_threads(int threads) {
    int n = threads > 0 ? threads : omp_get_num_procs();
    // IPCFP_ORACLE_MAX_THREADS caps "every processor" (the test-suite keeps the oracle to a few dozen threads)
    static const int cap = [] {
        const char* e = std::getenv("IPCFP_ORACLE_MAX_THREADS");
        return e ? std::atoi(e) : 0;
    }();
    if (threads <= 0 && cap > 0 && n > cap) n = cap;
    omp_set_num_threads(n);  // sticky: every parallel entry point sets it explicitly
    static int grown_for = 0;
    if (n > 1 && n > grown_for) {  // worker threads persist in libgomp's pool: grow each arena once
        pregrow_arenas();
        grown_for = n;
    }
    return n;
}

namespace {
// one subtree of the frontier, or the failure met where it would have been loaded
struct Sub {
    std::unique_ptr<Node> node;
    uint64_t height = 0, base = 0;
    bool failed = false;
    uint8_t status = 0;
    std::string what;
    std::vector<AmtItem> items;
};
}  // namespace

void amt_collect(const Blockstore& bs, const AmtRoot& root, const ValueChecker& check, std::vector<AmtItem>& out,
                 int threads) {
    out.clear();
    const int nthreads = use_threads(threads);
    std::vector<Sub> frontier(1);
    frontier[0].node.reset(new Node(root_node(root, check)));
    frontier[0].height = root.height;
    frontier[0].base = 0;
    // expand level by level until there is enough independent work; a child that fails to load stays in the
    // frontier as a failure AT ITS POSITION, so the first failure in depth-first order can be told at the end
    while (frontier.size() < size_t(nthreads) * 8) {
        bool any_link = false;
        for (const Sub& s : frontier) any_link = any_link || (!s.failed && s.node->is_link);
        if (!any_link) break;
        std::vector<Sub> next;
        for (Sub& s : frontier) {
            if (s.failed || !s.node->is_link) {
                next.push_back(std::move(s));
                continue;
            }
            if (s.height == 0) {
                Sub f;
                f.failed = true;
                f.status = IPCFP_ST_ERR_DECODE;
                f.what = "AMT link node at height 0";
                next.push_back(std::move(f));
                continue;
            }
            const uint64_t span = nodes_for_height(root.bit_width, s.height);
            uint32_t k = 0;
            for (uint32_t i = 0; i < s.node->width; ++i) {
                if (!s.node->bit(i)) continue;
                Sub c;
                c.height = s.height - 1;
                c.base = s.base + uint64_t(i) * span;
                try {
                    c.node.reset(new Node(load_child(bs, s.node->links[k], root.bit_width, check)));
                } catch (const Err& e) {
                    c.failed = true;
                    c.status = e.status;
                    c.what = e.what();
                }
                ++k;
                next.push_back(std::move(c));
            }
        }
        frontier.swap(next);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < int64_t(frontier.size()); ++q) {
        Sub& s = frontier[size_t(q)];
        if (s.failed) continue;
        try {
            node_for_each(bs, *s.node, s.height, root.bit_width, s.base, check,
                          [&](uint64_t i, const ValueLoc& v) { s.items.push_back(AmtItem{i, v}); });
        } catch (const Err& e) {
            s.failed = true;
            s.status = e.status;
            s.what = e.what();
        }
    }
    size_t total = 0;
    for (const Sub& s : frontier) {
        if (s.failed) throw Err(s.status, s.what);
        total += s.items.size();
    }
    out.reserve(total);
    for (const Sub& s : frontier) out.insert(out.end(), s.items.begin(), s.items.end());
}

}  // namespace orc
