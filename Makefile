# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: all lib oracle host test-cpu clean
all: lib oracle host
lib:
	$(MAKE) -C eigenkernel_amd/csrc -j4
oracle:
	$(MAKE) -C oracle
host: lib
	@if [ -x /opt/rocm/lib/llvm/bin/flang ]; then $(MAKE) -C host; else echo "flang missing: host skipped"; fi
test-cpu: all
	python -m pytest tests -x -q -m "not gpu"
clean:
	$(MAKE) -C eigenkernel_amd/csrc clean; $(MAKE) -C oracle clean; $(MAKE) -C host clean
