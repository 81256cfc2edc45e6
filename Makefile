# Convenience targets; everything is plain Python / hipcc underneath.
.PHONY: build test test-gpu bench profile clean
build:            ## hipcc -> usip_amd/libusip_hip.so, gcc -> oracle, g++ -> oracle/_ref (when /root/reference exists)
	python __graft_entry__.py
test:             ## CPU suite: oracle vs golden vectors, C ABI, host logic, gloo data parallel
	python -m pytest tests -q -m "not gpu"
test-gpu:         ## parity of every HIP operator and of the whole step (needs an MI355X)
	python -m pytest tests -q -m gpu
bench:            ## one JSON line: point-clouds/s, roofline of the dominant kernel, CPU baseline
	python bench.py
profile:          ## rocprofv3 kernel trace + HBM traffic counters -> gpurun_out/prof_<tag>/
	bash tools/profile_roofline.sh local
clean:
	rm -rf usip_amd/build usip_amd/libusip_hip.so oracle/libusip_oracle.so oracle/_ref
