import sys
sys.path.insert(0, '.')
import cholesky_amd as ca
for spec, worlds in [((40, 40, 40, 6, 64), (4, 8)), ((60, 60, 60, 8, 64), (8,)), ((100, 100, 100, 10, 64), (8,))]:
    plan = ca.Problem(*spec).plan()
    for world in worlds:
        for elem in (8, 4):
            out = []
            for r in range(world):
                dev = ca.Device(plan, 0)
                dev.set_partition(r, world)
                a = dev.alloc_arena(elem)
                out.append(a.backed_bytes)
                a.free()
                del dev
            full = plan.arena_doubles * elem
            print(spec, 'world', world, 'elem', elem, 'full GB %.2f' % (full / 1e9), 'rank 0 %.2f' % (out[0] / 1e9), 'other ranks GB min %.2f max %.2f' % (min(out[1:]) / 1e9, max(out[1:]) / 1e9), flush=True)
