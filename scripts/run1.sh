cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_abi_shim.py tests/test_gpu_full_size.py "tests/test_gpu_lzss.py::test_long_candidates_where_the_end_of_the_stream_binds" tests/test_gpu_huffman_decode.py -m gpu -x -q --durations=12 2>&1 | tail -40) > gpurun_out/t1.log 2>&1
tail -30 gpurun_out/t1.log
(timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err; echo rc=$?) 
tail -c 1500 gpurun_out/bench_r03a.json
