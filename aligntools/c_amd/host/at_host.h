/* at_host.h -- internals of the C host (CLI + reference-shaped wrappers). */
#ifndef AT_HOST_H
#define AT_HOST_H
#include "../../../include/aligntools.h"
#include "../../../include/aligntools_hip.h"

int at_parse_sites(const char *comment, int **pos_out);
at_handle *at_host_handle(void);   /* process-wide handle, created on first use; dies without a GPU */

#endif
