// fq_reframe.h - the reference's line reader, as a cut list.
//
// fastq_read_entry (reference src/fastq.c:245-261) reads a record with four gzgets() calls whose buffers hold
// MAX_LABEL_LENGTH = 1000 bytes for the two header lines and MAX_READ_LENGTH = 2 500 000 for sequence and quality
// (src/fastq.h:30-41): gzgets returns at most limit - 1 bytes, so a longer line comes back in pieces, and since the
// NEXT piece is read by the NEXT call - the one for the next field of the record - every later "line" of the file is
// out of step.  That is deterministic behaviour of the reference, and the drop-in reproduces it by handing the GPU an
// image in which the cuts are lines:
//
//     a piece that gzgets returns WITHOUT a newline (limit - 1 bytes of a longer line) is followed by "\0\n"
//
// - the NUL is the byte the reference's buffer holds behind the piece (its strings end there: strlen, %s, the scans
// of fastq_validate_entry all stop at it, and the library treats a NUL inside a line the same way), the '\n' only
// frames.  Everything else passes through unchanged.  Such an image is validated with FQG_VALIDATE_REFRAMED.
//
// Host work on purpose: it is a serial dependence (which limit applies to a line depends on every cut before it), it
// concerns input that real files do not contain, and it only cuts - nothing is validated here.
#pragma once
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fqg_codes.h"

namespace fqhost {

struct Reframer {
  unsigned phase = 0;  // which of a record's four gzgets calls reads next (0, 2: headers; 1, 3: sequence, quality)

  static size_t room(unsigned ph) { return (size_t)((ph & 1u) ? FQG_MAX_READ_LENGTH : FQG_MAX_LABEL_LENGTH) - 1; }

  // Cuts raw[0, n) - bytes of the file that follow what earlier calls took - the way the reference's reads would.
  // Returns how many raw bytes were taken: all of them when `at_end` (nothing follows), otherwise everything but a
  // trailing piece that is neither ended by a newline nor as long as its call's limit (it is offered again, with more
  // bytes behind it).  *clean = no cut fell into the taken bytes: they are their own re-framed form and `out` is left
  // alone; otherwise `out` holds the re-framed form of the taken bytes.  `cuts` (optional) collects where in `out`
  // the two bytes of every cut lie.
  size_t run(const char* raw, size_t n, bool at_end, std::string& out, bool* clean, std::vector<size_t>* cuts = nullptr) {
    size_t pos = 0, flushed = 0;
    bool cut = false;
    unsigned ph = phase;
    while (pos < n) {
      const size_t lim = room(ph), span = lim < n - pos ? lim : n - pos;
      const char* nl = static_cast<const char*>(memchr(raw + pos, '\n', span));
      if (nl) {  // the call returns the line with its newline
        pos = (size_t)(nl - raw) + 1;
        ph = (ph + 1) & 3u;
        continue;
      }
      if (span == lim) {  // limit - 1 bytes and no newline among them: the call returns them, the next call goes on
        if (!cut) {
          cut = true;
          out.clear();
        }
        out.append(raw + flushed, pos + lim - flushed);
        if (cuts) cuts->push_back(out.size());
        out.append("\0\n", 2);
        pos += lim;
        flushed = pos;
        ph = (ph + 1) & 3u;
        continue;
      }
      if (at_end) pos = n;  // the unterminated last line of the file: returned as it is
      break;
    }
    phase = ph;
    if (cut) out.append(raw + flushed, pos - flushed);
    *clean = !cut;
    return pos;
  }
};

}  // namespace fqhost
