// flatfile.h -- named flat arrays between tests/*.py (numpy) and the host smoke programs: int32 count, then per record
// char name[24], int32 kind (0 int32, 1 float32, 2 uint8), int32 n, n elements.  Test plumbing only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct FlatFile {
    std::map<std::string, std::vector<int32_t>> i;
    std::map<std::string, std::vector<float>> f;
    std::map<std::string, std::vector<uint8_t>> u;
    bool load(const char *path)
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) return false;
        int32_t nrec = 0;
        if (fread(&nrec, 4, 1, fp) != 1) { fclose(fp); return false; }
        for (int r = 0; r < nrec; r++) {
            char name[24]; int32_t kind, n;
            if (fread(name, 1, 24, fp) != 24 || fread(&kind, 4, 1, fp) != 1 || fread(&n, 4, 1, fp) != 1) { fclose(fp); return false; }
            name[23] = 0;
            const size_t esz = kind == 2 ? 1 : 4;
            std::vector<uint8_t> raw((size_t)n * esz);
            if (n && fread(raw.data(), esz, n, fp) != (size_t)n) { fclose(fp); return false; }
            if (kind == 0) { i[name].resize(n); if (n) memcpy(i[name].data(), raw.data(), raw.size()); }
            else if (kind == 1) { f[name].resize(n); if (n) memcpy(f[name].data(), raw.data(), raw.size()); }
            else u[name] = raw;
        }
        fclose(fp);
        return true;
    }
    const std::vector<int32_t> &I(const char *k) const { auto it = i.find(k); if (it == i.end()) { fprintf(stderr, "flatfile: no int array %s\n", k); exit(2); } return it->second; }
    const std::vector<float> &F(const char *k) const { auto it = f.find(k); if (it == f.end()) { fprintf(stderr, "flatfile: no float array %s\n", k); exit(2); } return it->second; }
    const std::vector<uint8_t> &U(const char *k) const { auto it = u.find(k); if (it == u.end()) { fprintf(stderr, "flatfile: no byte array %s\n", k); exit(2); } return it->second; }
    bool has(const char *k) const { return i.count(k) || f.count(k) || u.count(k); }
};

struct FlatWriter {
    FILE *fp; int32_t nrec; 
    explicit FlatWriter(const char *path) : fp(fopen(path, "wb")), nrec(0) { if (fp) fwrite(&nrec, 4, 1, fp); }
    ~FlatWriter() { if (fp) { fseek(fp, 0, SEEK_SET); fwrite(&nrec, 4, 1, fp); fclose(fp); } }
    void rec(const char *name, int32_t kind, int32_t n, const void *data)
    {
        char nm[24]; memset(nm, 0, 24); strncpy(nm, name, 23);
        fwrite(nm, 1, 24, fp); fwrite(&kind, 4, 1, fp); fwrite(&n, 4, 1, fp);
        if (n) fwrite(data, kind == 2 ? 1 : 4, n, fp);
        nrec++;
    }
    void ints(const char *name, const std::vector<int32_t> &v) { rec(name, 0, (int32_t)v.size(), v.data()); }
    void floats(const char *name, const std::vector<float> &v) { rec(name, 1, (int32_t)v.size(), v.data()); }
    void one(const char *name, int32_t v) { rec(name, 0, 1, &v); }
};
