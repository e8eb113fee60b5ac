#!/bin/bash
set -e
: > gpurun_out/lab_consumer.txt
for pol in plain nt sc1 sc0sc1 sc1nt; do
  RAFTCORR_LIB=$PWD/tools/lab_libs/$pol/libraftcorr_hip.so timeout -k 10 200 python tools/lab_consumer.py $pol >> gpurun_out/lab_consumer.txt 2>&1
done
