"""Run the step's GEMM shapes with the default tile choice (target of rocprofv3 --pmc passes), or summarise the passes.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d out -o a -- python tools/gemm_pmc.py
    python tools/gemm_pmc.py --summarise a_results.db b_results.db ...
"""
import importlib
import os
import sqlite3
import sys

SHAPES = [  # tag, ta, tb, M, N, K
    ('fwd  [E,F]x[F,F]', 0, 1, 19400, 200, 200),
    ('fwd  [N,4F]x[4F,F]', 0, 1, 8320, 200, 800),
    ('dgrad[N,F]x[F,4F]', 0, 0, 8320, 800, 200),
    ('wgrad[F,F] K=E', 1, 0, 200, 200, 19400),
    ('wgrad[F,F] K=N', 1, 0, 200, 200, 8320),
]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = importlib.import_module('3dinfomax_amd._lib').load()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    for _, ta, tb, M, N, K in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        for _ in range(5):
            assert lib.i3d_gemm_f32(ta, tb, M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), N, None, 0, st) == 0
    torch.cuda.synchronize()


def summarise(dbs):
    rows = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*) from counters_collection "
             "where kernel_name like '%gemm_f32%' group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name")
        for name, gx, gy, gz, cname, val, n in cur.execute(q):
            short = name[name.index('Shape<'):name.index('>', name.index('Shape<')) + 1] + f' grid {gx // 256}x{gy}x{gz}'
            rows.setdefault(short, {})[cname] = val
    for k, d in rows.items():
        print(k)
        for c in sorted(d):
            print(f'    {c:28s} {d[c]:16.1f}')
        wc = d.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU'):
                if c in d:
                    print(f'    {c + " / SQ_WAVE_CYCLES":40s} {d[c] / wc:8.3f}')
        if 'SQ_BUSY_CU_CYCLES' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
            print(f'    MFMA busy / CU busy cycles               {d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CU_CYCLES"]:8.3f}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2:])
    else:
        run()
