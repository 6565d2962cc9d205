run() { python tools/kbench.py --case "a:xyz;;dft;;printable;;auto" --case "(a|b)*c:x;;nft;;printable;;auto" --case "[a-z]+ing:X;;dft;;printable;;auto" --case " +: ;;nft;;printable;;auto" --steps 5 2>&1 | grep "pattern\|void"; }
TRRE_TRACE_VOID=1 TRRE_G16_SPLICE=1 run
run
