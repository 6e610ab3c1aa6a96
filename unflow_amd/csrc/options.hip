// unflow_set_option / unflow_get_option: the only way the library's process-wide options (options.h) change.
#include <cstring>
#include "common.h"
#include "options.h"

namespace unflow {
Options& options() {
  static Options o;
  return o;
}
}  // namespace unflow

namespace {
struct Entry {
  const char* name;
  int unflow::Options::*field;
};
const Entry kTable[] = {
#define X(name, def) {#name, &unflow::Options::name},
    UNFLOW_OPTION_LIST(X)
#undef X
};
}  // namespace

UNFLOW_API int unflow_set_option(const char* name, int value) {
  if (!name) return UNFLOW_ERR_NULL;
  for (const Entry& e : kTable)
    if (!strcmp(e.name, name)) {
      unflow::options().*(e.field) = value;
      return UNFLOW_OK;
    }
  return UNFLOW_ERR_UNSUPPORTED;
}

UNFLOW_API int unflow_get_option(const char* name, int* value) {
  if (!name || !value) return UNFLOW_ERR_NULL;
  for (const Entry& e : kTable)
    if (!strcmp(e.name, name)) {
      *value = unflow::options().*(e.field);
      return UNFLOW_OK;
    }
  return UNFLOW_ERR_UNSUPPORTED;
}

// option names, '\n'-separated (for the host layer's environment bridge and for tests)
UNFLOW_API const char* unflow_option_names(void) {
  return
#define X(name, def) #name "\n"
      UNFLOW_OPTION_LIST(X)
#undef X
      ;
}
