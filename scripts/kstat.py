"""Average duration (us) of the kernels whose name contains argv[2], from a rocprofv3 kernel_stats.csv (argv[1])."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        name = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        print("%s calls %s avg %.1f us" % (name, r['Calls'], float(r['AverageNs']) / 1e3))
