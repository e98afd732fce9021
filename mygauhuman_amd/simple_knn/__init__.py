"""`simple_knn` drop-in (SK/ext.cpp:15-17): `from simple_knn._C import distCUDA2`."""
