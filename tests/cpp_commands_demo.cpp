// Drives vectordb_amd/cpp/vdb_commands.hpp the way the reference's CommandProcessor would:
// argument vectors in, serialized replies out.  The serializer below has the static interface of the reference's
// ResponseSerializer (response_serializer.hpp:31-58) and writes text, so that the pytest can read the replies.
//   argv[1] = "host": parsing, errors, VADD / VCARD / VDEL (no GPU);  "gpu": VSEARCH as well.
#include <cstdio>
#include <cstring>
#include <random>
#include <string_view>

#include "../vectordb_amd/cpp/vdb_commands.hpp"

struct TextSerializer {
    static void put(std::vector<uint8_t>& b, const std::string& s) { b.insert(b.end(), s.begin(), s.end()); }
    static void serialize_nil(std::vector<uint8_t>& b) { put(b, "nil\n"); }
    static void serialize_error(std::vector<uint8_t>& b, int32_t code, std::string_view msg) {
        put(b, "err " + std::to_string(code) + " " + std::string(msg));
    }
    static void serialize_string(std::vector<uint8_t>& b, std::string_view s) { put(b, "str " + std::string(s) + "\n"); }
    static void serialize_integer(std::vector<uint8_t>& b, int64_t v) { put(b, "int " + std::to_string(v) + "\n"); }
    static void serialize_double(std::vector<uint8_t>& b, double v) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "dbl %a\n", v);
        put(b, buf);
    }
};

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    vdb_amd::VectorCommands<TextSerializer> vectors;
    vectors.initial_capacity = 64;  // grows through resizeIndex
    auto run = [&](std::vector<std::string> args) {
        std::vector<uint8_t> out;
        const std::string cmd = args[0];
        if (cmd == "vadd") vectors.vadd(args, out);
        else if (cmd == "vsearch") vectors.vsearch(args, out);
        else if (cmd == "vdel") vectors.vdel(args, out);
        else vectors.vcard(args, out);
        std::printf("> %s\n%.*s", cmd.c_str(), (int)out.size(), (const char*)out.data());
    };
    const size_t n = 300, dim = 8;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<std::vector<std::string>> rows;
    for (size_t i = 0; i < n; ++i) {
        std::vector<std::string> a = {"vadd", "docs", std::to_string(1000 + i)};
        for (size_t c = 0; c < dim; ++c) {
            char buf[32];
            std::snprintf(buf, sizeof buf, "%.9g", u(rng));
            a.push_back(buf);
        }
        rows.push_back(a);
    }
    for (auto& a : rows) run(a);
    run({"vcard", "docs"});
    run({"vcard", "nothing"});
    run({"vadd", "docs", "5", "0.5"});                 // wrong dimension
    run({"vadd", "docs", "x", "1", "2"});              // bad label
    run({"vadd", "docs", "7", "1", "2", "zebra", "4", "5", "6", "7", "8"});
    run({"vdel", "docs", "1003"});
    run({"vdel", "docs", "999999"});
    run({"vsearch", "nothing", "3", "1", "2"});
    if (gpu) {
        for (size_t q = 0; q < 5; ++q) {
            std::vector<std::string> a = {"vsearch", "docs", "4"};
            a.insert(a.end(), rows[q].begin() + 3, rows[q].end());
            a.push_back("EF");
            a.push_back("50");
            run(a);
        }
    }
    return 0;
}
