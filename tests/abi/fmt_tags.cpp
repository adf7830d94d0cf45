// fmt_tags.cpp -- prints floats the two ways RawAlign's PAF tags do (test infrastructure):
//   aln:s: differences through `std::stringstream << float`   (dtwresult_to_string, src/rmap.cpp:580-592)
//   alns:f: and the other :f: tags through std::to_string      (src/rmap.cpp:731-742)
// stdin: one 32-bit pattern (hex) per line; stdout: "<ostream form> <to_string form>" per line.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

int main()
{
    unsigned u;
    while (scanf("%x", &u) == 1) {
        float f;
        memcpy(&f, &u, 4);
        std::stringstream ss;
        ss << f;
        std::cout << ss.str() << " " << std::to_string(f) << "\n";
    }
    return 0;
}
