/* oracle/ko_run.c — TEST INFRASTRUCTURE ONLY: command-line scenario runner for the C restatement. */
#include "klang_oracle.h"
#include <stdio.h>
int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	const int rc = ko_run_scenario(argv[1], argv[2]);
	if (rc) fprintf(stderr, "ko_run_scenario failed: %d\n", rc);
	return rc;
}
