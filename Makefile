# top-level convenience targets
all: lib oracle
lib:
	$(MAKE) -C outerspace_amd/csrc all
oracle:
	$(MAKE) -C oracle all
test:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu
bench:
	python bench.py
clean:
	$(MAKE) -C outerspace_amd/csrc clean
	$(MAKE) -C oracle clean
.PHONY: all lib oracle test test-gpu bench clean
