"""A digest of a built tree that does not depend on how the nodes are numbered or which child sits in which slot: leaves hash
their sorted triangle ids, inner nodes the unordered pair of (child box bits, child digest).  Two builders produce THE SAME
TREE iff the digests agree (tests/test_gpu_bvhbuild.py, tools/tree_digest.py)."""
import hashlib

import numpy as np


def digest(nodes, tris):
    """nodes: [n, 16] uint32 (rtbvh::Node: lo0 hi0 lo1 hi1 child[2] pad[2]); tris: [m, 12] uint32 (TriRec: word 9 = id)."""
    n = len(nodes)
    ids = tris[:, 9]
    h = [None] * n
    child = nodes[:, 12:14].astype(np.int32)
    box = [nodes[:, 0:6].tobytes(), nodes[:, 6:12].tobytes()]
    for i in range(n - 1, -1, -1):  # (children are numbered behind their parents by every builder)
        parts = []
        for c in range(2):
            ch = int(child[i, c])
            if ch < 0:
                code = (~ch) & 0xFFFFFFFF
                first, cnt = code >> 3, (code & 7) + 1
                hc = hashlib.blake2b(np.sort(ids[first:first + cnt]).tobytes(), digest_size=12).digest()
            else:
                assert ch > i
                hc = h[ch]
                h[ch] = None
            parts.append(box[c][24 * i:24 * i + 24] + hc)
        parts.sort()
        h[i] = hashlib.blake2b(parts[0] + parts[1], digest_size=12).digest()
    return h[0].hex()


def context_digest(ctx):
    nodes, tris = ctx.bvh_export()
    return digest(np.ascontiguousarray(nodes).view(np.uint32).reshape(len(nodes), 16), np.ascontiguousarray(tris).view(np.uint32).reshape(len(tris), 12))


def first_differences(A, B, limit=5):
    """Walks two exported trees together from their roots, pairing children by their boxes; prints where they part."""
    (na, ta), (nb, tb) = A, B
    found = 0
    stack = [(0, 0, 0)]
    while stack and found < limit:
        i, j, depth = stack.pop()
        ca = [(na[i, 0:6].tobytes(), int(np.int32(na[i, 12]))), (na[i, 6:12].tobytes(), int(np.int32(na[i, 13])))]
        cb = [(nb[j, 0:6].tobytes(), int(np.int32(nb[j, 12]))), (nb[j, 6:12].tobytes(), int(np.int32(nb[j, 13])))]
        if sorted(x[0] for x in ca) != sorted(x[0] for x in cb):
            found += 1
            print("  depth %d: child boxes differ" % depth)
            for name, c, nn in (("A", ca, na), ("B", cb, nb)):
                for bx, ref in c:
                    print("    %s %s ref %d" % (name, np.frombuffer(bx, np.float32), ref))
            continue
        if ca[0][0] != cb[0][0]:
            cb.reverse()
        for (bx, ra), (_, rb) in zip(ca, cb):
            if (ra < 0) != (rb < 0):
                found += 1
                print("  depth %d: leaf against inner node under equal boxes" % (depth + 1))
            elif ra < 0:
                ia = np.sort(ta[(~ra & 0xFFFFFFFF) >> 3:((~ra & 0xFFFFFFFF) >> 3) + ((~ra) & 7) + 1, 9])
                ib = np.sort(tb[(~rb & 0xFFFFFFFF) >> 3:((~rb & 0xFFFFFFFF) >> 3) + ((~rb) & 7) + 1, 9])
                if not np.array_equal(ia, ib):
                    found += 1
                    print("  depth %d: leaves differ: %s | %s" % (depth + 1, ia, ib))
            else:
                stack.append((ra, rb, depth + 1))
    return found
