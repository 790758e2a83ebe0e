/* Writes tests/golden/hdf5/chunked.hdf5: the four BubbleML field names as small (7, 10, 12) float32 datasets in the storage layouts
 * h5py produces when chunking / compression is asked for (old-style file, chunked layout, B-tree v1 chunk index):
 *   dfun         chunked (3, 4, 5), no filter (edge chunks stick out of the dataset on every axis)
 *   temperature  chunked + gzip(4)
 *   velx         chunked + shuffle + gzip(6)
 *   vely         chunked + gzip(1) + fletcher32
 *   counts       int16 (5, 6), chunked (2, 4) + shuffle + gzip
 * value[t][y][x] = (131 t + 17 y + 3 x + 1000 field) / 4  (exact in float32, so the test recomputes it).
 * Build and run (this container ships libhdf5 1.10.6 under /opt/conda):  h5cc make_chunked_fixture.c -o /tmp/mk && /tmp/mk chunked.hdf5 */
#include <hdf5.h>
#include <stdio.h>
#include <stdint.h>

int main(int argc, char** argv) {
    const char* names[4] = {"dfun", "temperature", "velx", "vely"};
    hsize_t dims[3] = {7, 10, 12}, chunk[3] = {3, 4, 5};
    static float buf[7][10][12];
    hid_t f = H5Fcreate(argc > 1 ? argv[1] : "chunked.hdf5", H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    for (int k = 0; k < 4; ++k) {
        for (int t = 0; t < 7; ++t)
            for (int y = 0; y < 10; ++y)
                for (int x = 0; x < 12; ++x) buf[t][y][x] = (float)(131 * t + 17 * y + 3 * x + 1000 * k) * 0.25f;
        hid_t sp = H5Screate_simple(3, dims, NULL);
        hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_chunk(pl, 3, chunk);
        if (k == 1) H5Pset_deflate(pl, 4);
        if (k == 2) { H5Pset_shuffle(pl); H5Pset_deflate(pl, 6); }
        if (k == 3) { H5Pset_deflate(pl, 1); H5Pset_fletcher32(pl); }
        hid_t d = H5Dcreate2(f, names[k], H5T_IEEE_F32LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
        H5Dclose(d); H5Pclose(pl); H5Sclose(sp);
    }
    {
        hsize_t d2[2] = {5, 6}, c2[2] = {2, 4};
        int16_t v[5][6];
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 6; ++x) v[y][x] = (int16_t)(100 * y - 7 * x);
        hid_t sp = H5Screate_simple(2, d2, NULL);
        hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_chunk(pl, 2, c2); H5Pset_shuffle(pl); H5Pset_deflate(pl, 4);
        hid_t d = H5Dcreate2(f, "counts", H5T_STD_I16LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_INT16, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
        H5Dclose(d); H5Pclose(pl); H5Sclose(sp);
    }
    H5Fclose(f);
    return 0;
}
