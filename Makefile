sanitize:
	bash tools/sanitize.sh
.PHONY: sanitize
