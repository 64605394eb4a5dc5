/* at_host.h -- internals of the C host (CLI + reference-shaped wrappers). */
#ifndef AT_HOST_H
#define AT_HOST_H
#include "../../../include/aligntools.h"
#include "../../../include/aligntools_hip.h"

int at_parse_sites(const char *comment, int **pos_out);
/* malloc / realloc / strdup that die() with the reference's mycalloc message (alignment.h:81-87) instead of returning NULL */
void *at_xmalloc(size_t n);
void *at_xrealloc(void *p, size_t n);
char *at_xstrdup(const char *s);
at_handle *at_host_handle(void);   /* process-wide handle, created on first use; dies without a GPU */

#endif
