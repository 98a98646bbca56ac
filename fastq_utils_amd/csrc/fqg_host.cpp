// fqg_host.cpp - host-only parts of the C-ABI: the once-per-file probes.
#include <regex.h>

#include <string>

#include "../../include/fqg.h"

namespace {
bool rx_match(const char* pattern, int cflags, const char* s) {
  regex_t rx;
  if (regcomp(&rx, pattern, cflags) != 0) return false;
  const bool hit = regexec(&rx, s, 0, nullptr, 0) == 0;
  regfree(&rx);
  return hit;
}
}  // namespace

extern "C" {

// The decision ladder of fastq_get_readname (reference src/fastq.c:459-478) with the three
// POSIX patterns of src/fastq.c:672,694,714.  The patterns are the specification of the
// formats, so they are used as they are.
int fqg_probe_readname_format(const char* s) {
  if (!s) return FQG_NAME_UNDEF;
  if (rx_match("[A-Z0-9:]* [1234]:[YN]:[0-9]*.*", 0, s)) return FQG_NAME_CASAVA18;
  if (rx_match("^[0-9]+[\n\r]?$", REG_EXTENDED, s)) return FQG_NAME_INTEGER;
  // a name WITHOUT a "/1"-like suffix is kept whole (NOP shares INTEGERNAME's value)
  if (!rx_match("[# \t/:][0-9abAB][\n\r]?$", REG_EXTENDED, s)) return FQG_NAME_NOP;
  return FQG_NAME_DEFAULT;
}

// is_color_space, reference src/fastq.c:731-754
int fqg_probe_space(const char* seq_line) {
  if (!seq_line) return FQG_SPACE_UNDEF;
  return rx_match("^[GT]?[0123n\\.NtT]+\n?$", REG_EXTENDED, seq_line) ? FQG_SPACE_COLOUR : FQG_SPACE_SEQ;
}

int fqg_probe_first_record(const void* host_image, uint64_t nbytes, int is_pe, fqg_file_state* out) {
  if (!host_image || !out) return FQG_ERR_ARG;
  const char* b = static_cast<const char*>(host_image);
  // the first two lines as gzgets would hand them over (limits of src/fastq.c:249,251)
  auto take_line = [&](uint64_t& pos, uint64_t limit) {
    std::string s;
    while (pos < nbytes && s.size() < limit - 1) {
      const char ch = b[pos++];
      s.push_back(ch);
      if (ch == '\n') break;
    }
    return s;
  };
  uint64_t pos = 0;
  const std::string hdr = take_line(pos, FQG_MAX_LABEL_LENGTH);
  const std::string seq = take_line(pos, FQG_MAX_READ_LENGTH);
  out->is_pe = is_pe;
  out->reserved = 0;
  out->readname_format = FQG_NAME_UNDEF;
  out->space = FQG_SPACE_UNDEF;
  if (hdr.empty() || seq.empty()) return FQG_ERR_ARG;
  // c_str() cuts at an embedded NUL exactly like the reference's C strings
  out->readname_format = fqg_probe_readname_format(hdr.c_str() + 1);
  out->space = fqg_probe_space(seq.c_str());
  return 0;
}

}  // extern "C"
